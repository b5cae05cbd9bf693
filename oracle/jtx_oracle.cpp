/*
 * jtx_oracle.cpp -- CPU ORACLE: a scalar restatement of the JTX-PathTracer hot path
 * (BVH build + traversal, Moller-Trumbore, lights, BxDFs, integrateMIS, film).
 *
 * TEST INFRASTRUCTURE ONLY (see jtx_oracle.h).  PARITY UNPINNED (see jtx_oracle.h):
 * every function cites the reference file:line it restates; the semantics of the
 * un-vendored jtx:: math are the ones fixed in DESIGN.md "jtx math spec".
 *
 * Build: g++ -O2 -ffp-contract=off -fno-fast-math -fopenmp (oracle/Makefile).  All
 * arithmetic is IEEE fp32, one rounding per written operation, evaluated in the
 * order written -- the HIP kernels are required to match this bit for bit.
 */
#include "jtx_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <vector>
#include <omp.h>

namespace {

// ---------------------------------------------------------------------------------------------
// jtx math spec: Vec3 {x,y,z}; all ops componentwise, left to right.
// ---------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct V2 { float x, y; };

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3(float s) { return V3{s, s, s}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 operator/(float s, V3 a) { return {s / a.x, s / a.y, s / a.z}; }
inline V3 operator-(float s, V3 a) { return {s - a.x, s - a.y, s - a.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float absdot(V3 a, V3 b) { return std::fabs(dot(a, b)); }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float lenSqr(V3 a) { return dot(a, a); }
inline float len(V3 a) { return std::sqrt(lenSqr(a)); }
inline V3 normalize(V3 a) { return a / len(a); }
inline bool nonzero(V3 a) { return a.x != 0 || a.y != 0 || a.z != 0; } // Vec3::operator bool
inline V3 vmin(V3 a, V3 b) { return {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z}; }
inline V3 vmax(V3 a, V3 b) { return {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z}; }
inline float fmax2(float a, float b) { return a > b ? a : b; }   // jtx::max
inline float fmin2(float a, float b) { return a < b ? a : b; }   // jtx::min
inline float sqr(float x) { return x * x; }
inline float safeSqrt(float x) { return std::sqrt(fmax2(0.0f, x)); }
inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline float lerpf(float a, float b, float t) { return (1 - t) * a + t * b; }           // lerp(a,b,t)
inline V3 lerp3(V3 a, V3 b, float t) { return (1 - t) * a + t * b; }
inline V3 faceForward(V3 n, V3 v) { return dot(n, v) < 0 ? -n : n; }
inline bool sameHemisphere(V3 a, V3 b) { return a.z * b.z > 0; }
inline float cosTheta(V3 w) { return w.z; }
inline float absCosTheta(V3 w) { return std::fabs(w.z); }
inline float cos2Theta(V3 w) { return w.z * w.z; }
inline float sin2Theta(V3 w) { return fmax2(0.0f, 1 - cos2Theta(w)); }
inline float sinTheta(V3 w) { return std::sqrt(sin2Theta(w)); }
inline float tan2Theta(V3 w) { return sin2Theta(w) / cos2Theta(w); }
inline float cosPhi(V3 w) { float s = sinTheta(w); return s == 0 ? 1.0f : clampf(w.x / s, -1, 1); }
inline float sinPhi(V3 w) { float s = sinTheta(w); return s == 0 ? 0.0f : clampf(w.y / s, -1, 1); }

const float PI_F      = 3.14159265358979323846f;
const float INV_PI    = 1.0f / PI_F;          // sampling.hpp:7
const float PI_OVER_4 = PI_F / 4;             // sampling.hpp:3
const float PI_OVER_2 = PI_F / 2;             // sampling.hpp:4
const float INF_F     = std::numeric_limits<float>::infinity();
const float RAY_EPSILON = 1e-4f;              // scene.hpp:9

// ---------------------------------------------------------------------------------------------
// Deterministic sin/cos (DESIGN.md "sincos spec"): k = nearest integer to x*(2/pi); three-term
// Cody-Waite reduction r = ((x - k*A) - k*B) - k*C; degree-7/8 polynomials on |r| <= pi/4.
// Each written operation is one fp32 rounding.  Valid (and tested) for |x| <= 16.
// ---------------------------------------------------------------------------------------------
std::atomic<int> g_sincos_mode{0};
std::atomic<uint64_t> *g_node_hist = nullptr;     // diagnostic: per-node visit counts (ora_node_histogram)

inline void det_sincos(float x, float *s_out, float *c_out) {
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float A = 1.5703125f;                 // pi/2 split: 8 significant bits
    const float B = 4.837512969970703125e-4f;   // next 11 bits
    const float C = 7.54978995489188216e-8f;    // remainder
    float q = x * TWO_OVER_PI;
    float kf = std::floor(q + 0.5f);
    int k = (int) kf;
    float r = ((x - kf * A) - kf * B) - kf * C;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    switch (k & 3) {
        case 0: *s_out = sp;  *c_out = cp;  break;
        case 1: *s_out = cp;  *c_out = -sp; break;
        case 2: *s_out = -sp; *c_out = -cp; break;
        default: *s_out = -cp; *c_out = sp; break;
    }
}
inline float jsin(float x) {
    if (g_sincos_mode.load(std::memory_order_relaxed) == 1) return std::sin(x);
    float s, c; det_sincos(x, &s, &c); return s;
}
inline float jcos(float x) {
    if (g_sincos_mode.load(std::memory_order_relaxed) == 1) return std::cos(x);
    float s, c; det_sincos(x, &s, &c); return c;
}

// ---------------------------------------------------------------------------------------------
// RNG: PCG RXS-M-XS-32 with the reference's ">> 2" output step (util/rand.hpp:42-107)
// ---------------------------------------------------------------------------------------------
inline uint32_t fnv1a_3(uint32_t x, uint32_t y, uint32_t n) {   // rand.hpp:13-24
    uint32_t h = 2166136261u;
    h ^= x; h *= 16777619u;
    h ^= y; h *= 16777619u;
    h ^= n; h *= 16777619u;
    return h;
}
struct Rng {
    uint32_t state;
    // RNG(x,y,n): state_ read as 0 before init; init = advance, += seed, advance (rand.hpp:93-97)
    Rng(uint32_t x, uint32_t y, uint32_t n) {
        state = 0;
        advance();
        state += fnv1a_3(x, y, n);
        advance();
    }
    uint32_t advance() {                                         // rand.hpp:99-104
        uint32_t s = state;
        state = state * 747796405u + 2891336453u;
        uint32_t word = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
        return (word >> 2u) ^ word;
    }
    float f() { return (advance() & 0xFFFFFF) / 16777216.0f; }   // rand.hpp:132-134
    // sampleRange(range) with t = (-range) % range == 0 in int arithmetic: exactly one advance,
    // result hi32(x*range); range==0 (one light) read as index 0 (rand.hpp:73-84)
    uint32_t sampleRange(int range) {
        uint32_t x = advance();
        if (range <= 0) return 0;
        return (uint32_t) ((uint64_t(x) * uint64_t(range)) >> 32);
    }
};

// ---------------------------------------------------------------------------------------------
// Sampling warps (sampling.hpp:22-63)
// ---------------------------------------------------------------------------------------------
inline V2 sampleUniformDiskPolar(V2 u) {                          // sampling.hpp:22-26
    float r = std::sqrt(u.x);
    float theta = 2 * PI_F * u.y;
    return {r * jcos(theta), r * jsin(theta)};
}
inline V2 sampleUniformDiskConcentric(V2 u) {                     // sampling.hpp:28-46
    V2 o = {2.0f * u.x - 1.0f, 2.0f * u.y - 1.0f};
    if (o.x == 0 && o.y == 0) return {0.0f, 0.0f};
    float r, theta;
    if (std::fabs(o.x) > std::fabs(o.y)) {
        r = o.x;
        theta = PI_OVER_4 * (o.y / o.x);
    } else {
        r = o.y;
        theta = PI_OVER_2 - PI_OVER_4 * (o.x / o.y);
    }
    return {r * jcos(theta), r * jsin(theta)};
}
inline V3 sampleCosineHemisphere(V2 u) {                          // sampling.hpp:56-59
    V2 d = sampleUniformDiskConcentric(u);
    return {d.x, d.y, safeSqrt(1 - d.x * d.x - d.y * d.y)};
}
inline float cosineHemispherePDF(float c) { return c * INV_PI; } // sampling.hpp:61-63

// ---------------------------------------------------------------------------------------------
// AABB (util/aabb.hpp) and Interval (util/interval.hpp)
// ---------------------------------------------------------------------------------------------
struct Box {
    V3 pmin, pmax;
    Box() {                                                       // aabb.hpp:11-16
        float lo = std::numeric_limits<float>::lowest(), hi = std::numeric_limits<float>::max();
        pmin = {hi, hi, hi}; pmax = {lo, lo, lo};
    }
    Box(V3 a, V3 b) { pmin = vmin(a, b); pmax = vmax(a, b); }     // aabb.hpp:18-21
    void expand(const Box &o) { pmin = vmin(pmin, o.pmin); pmax = vmax(pmax, o.pmax); }  // aabb.hpp:33-37
    void expand(V3 p) { pmin = vmin(pmin, p); pmax = vmax(pmax, p); }                    // aabb.hpp:39-43
    V3 diagonal() const { return pmax - pmin; }
    float surfaceArea() const { V3 d = diagonal(); return 2 * (d.x * d.y + d.x * d.z + d.y * d.z); } // aabb.hpp:87-90
    int longestAxis() const {                                     // aabb.hpp:51-56
        V3 d = diagonal();
        if (d.x > d.y && d.x > d.z) return 0;
        if (d.y > d.z) return 1;
        return 2;
    }
    V3 offset(V3 p) const {                                       // aabb.hpp:58-64
        V3 o = p - pmin;
        if (pmax.x > pmin.x) o.x /= pmax.x - pmin.x;
        if (pmax.y > pmin.y) o.y /= pmax.y - pmin.y;
        if (pmax.z > pmin.z) o.z /= pmax.z - pmin.z;
        return o;
    }
};

// AABB::hit (aabb.hpp:66-81): per-axis 1/d, swap when near>far, strict t0>t1 reject
inline bool aabbHit(const float pmin[3], const float pmax[3], V3 o, V3 d, float t0, float t1) {
    for (int i = 0; i < 3; ++i) {
        float invDir = 1 / d[i];
        float tNear = (pmin[i] - o[i]) * invDir;
        float tFar  = (pmax[i] - o[i]) * invDir;
        if (tNear > tFar) std::swap(tNear, tFar);
        t0 = tNear > t0 ? tNear : t0;
        t1 = tFar < t1 ? tFar : t1;
        if (t0 > t1) return false;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// Scene storage
// ---------------------------------------------------------------------------------------------
struct Mesh {
    int numTris, numVerts;
    std::vector<int32_t> indices;
    std::vector<V3> vertices, normals;
    std::vector<V2> uvs;
    bool hasUVs;
    int material;
    float m[4][4];
    V3 applyToPoint(V3 p) const {   // affine, no w divide
        return {m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3],
                m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3],
                m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3]};
    }
    V3 applyToNormal(V3 n) const {  // upper 3x3
        return {m[0][0] * n.x + m[0][1] * n.y + m[0][2] * n.z,
                m[1][0] * n.x + m[1][1] * n.y + m[1][2] * n.z,
                m[2][0] * n.x + m[2][1] * n.y + m[2][2] * n.z};
    }
    void getVertices(int index, V3 &v0, V3 &v1, V3 &v2) const {   // mesh.hpp:71-77
        const int32_t *i = &indices[3 * index];
        v0 = applyToPoint(vertices[i[0]]);
        v1 = applyToPoint(vertices[i[1]]);
        v2 = applyToPoint(vertices[i[2]]);
    }
    void getNormals(int index, V3 &n0, V3 &n1, V3 &n2) const {    // mesh.hpp:92-97
        const int32_t *i = &indices[3 * index];
        n0 = applyToNormal(normals[i[0]]);
        n1 = applyToNormal(normals[i[1]]);
        n2 = applyToNormal(normals[i[2]]);
    }
    void getUVs(int index, V2 &a, V2 &b, V2 &c) const {           // mesh.hpp:99-104 (+Q3: no uvs => 0)
        if (!hasUVs) { a = b = c = V2{0, 0}; return; }
        const int32_t *i = &indices[3 * index];
        a = uvs[i[0]]; b = uvs[i[1]]; c = uvs[i[2]];
    }
    Box tBounds(int index) const {                                // mesh.hpp:79-84
        V3 v0, v1, v2; getVertices(index, v0, v1, v2);
        Box b(v0, v1); b.expand(v2); return b;
    }
};

struct Tri {                     // Triangle mesh.hpp:202-210
    int index, meshIndex;
    Box bounds;
    V3 centroid() const { return 0.5f * bounds.pmin + 0.5f * bounds.pmax; }
};

struct Texture { int w, h, c; std::vector<float> data; };

struct Hit {                     // SurfaceIntersection material.hpp:25-40 (tangent/bitangent unused)
    V3 point, normal;
    V2 uv;
    int material;
    float t;
    bool frontFace;
    int prim; float b1, b2;      // extra outputs for per-ray parity tests
};

struct Counters {
    uint64_t n_camera = 0, n_closest = 0, n_any = 0, n_nodes_closest = 0, n_tri_closest = 0, n_accept = 0,
             n_nodes_any = 0, n_tri_any = 0, n_shade = 0;
    uint64_t n_shade_class[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n_eval_class[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void shade(int cls) { n_shade++; if (cls >= 0 && cls < 8) n_shade_class[cls]++; }
    void eval(int cls) { if (cls >= 0 && cls < 8) n_eval_class[cls]++; }
    void add(const Counters &o) {
        for (int i = 0; i < 8; ++i) { n_shade_class[i] += o.n_shade_class[i]; n_eval_class[i] += o.n_eval_class[i]; }
        n_camera += o.n_camera; n_closest += o.n_closest; n_any += o.n_any;
        n_nodes_closest += o.n_nodes_closest; n_tri_closest += o.n_tri_closest; n_accept += o.n_accept;
        n_nodes_any += o.n_nodes_any; n_tri_any += o.n_tri_any; n_shade += o.n_shade;
    }
};

} // namespace

struct ora_scene {
    std::vector<Mesh> meshes;
    std::vector<Tri> triangles;          // Scene::triangles
    std::vector<Tri> ordered;            // Scene::triangles_ after build
    std::vector<ora_bvh_node> nodes;     // Scene::nodes_
    std::vector<ora_material> materials;
    std::vector<ora_light> lights;
    std::vector<Texture> textures;
    V3 sky;
    int maxDepth = 0;
    float radius = 0;
};

namespace {

// ---------------------------------------------------------------------------------------------
// BVH build (bvh.cpp:9-133) + flatten (bvh.cpp:135-149)
// ---------------------------------------------------------------------------------------------
struct BuildNode {
    Box bbox;
    BuildNode *children[2] = {nullptr, nullptr};
    int splitAxis = 0, firstPrimOffset = 0, numPrimitives = 0;
};

BuildNode *makeLeaf(Tri *prims, size_t n, const Box &bounds, int *orderedOffset, std::vector<Tri> &ordered,
                    BuildNode *node) {
    int first = *orderedOffset;
    *orderedOffset += (int) n;
    for (size_t i = 0; i < n; ++i) ordered[first + i] = prims[i];
    node->firstPrimOffset = first;
    node->numPrimitives = (int) n;
    node->bbox = bounds;
    return node;
}

BuildNode *buildTree(Tri *prims, size_t n, int *totalNodes, int *orderedOffset, std::vector<Tri> &ordered,
                     int maxPrimsInNode) {
    BuildNode *node = new BuildNode();
    (*totalNodes)++;

    Box bounds;
    for (size_t i = 0; i < n; ++i) bounds.expand(prims[i].bounds);

    if (bounds.surfaceArea() == 0 || n == 1) return makeLeaf(prims, n, bounds, orderedOffset, ordered, node); // bvh.cpp:18-28

    Box centroidBounds;
    for (size_t i = 0; i < n; ++i) centroidBounds.expand(prims[i].centroid());
    int dim = centroidBounds.longestAxis();
    if (centroidBounds.pmin[dim] == centroidBounds.pmax[dim])                                              // bvh.cpp:36-46
        return makeLeaf(prims, n, bounds, orderedOffset, ordered, node);

    size_t mid = n / 2;
    if (n == 2) {                                                                                           // bvh.cpp:50-57
        std::nth_element(prims, prims + mid, prims + n,
                         [dim](const Tri &a, const Tri &b) { return a.centroid()[dim] < b.centroid()[dim]; });
    } else {
        const int NB = 12;                                                                                  // bvh.cpp:60
        struct Bucket { int count = 0; Box bounds; } buckets[NB];
        for (size_t i = 0; i < n; ++i) {
            int b = (int) (NB * centroidBounds.offset(prims[i].centroid())[dim]);
            if (b == NB) b = NB - 1;
            buckets[b].count++;
            buckets[b].bounds.expand(prims[i].bounds);
        }
        const int NS = NB - 1;
        float costs[NS] = {};
        int countBelow = 0; Box boundsBelow;
        for (int i = 0; i < NS; ++i) {                                                                      // bvh.cpp:74-81
            countBelow += buckets[i].count;
            boundsBelow.expand(buckets[i].bounds);
            costs[i] += countBelow * boundsBelow.surfaceArea();
        }
        int countAbove = 0; Box boundsAbove;
        for (int i = NB - 1; i > 0; --i) {                                                                  // bvh.cpp:84-90
            countAbove += buckets[i].count;
            boundsAbove.expand(buckets[i].bounds);
            costs[i - 1] += countAbove * boundsAbove.surfaceArea();
        }
        int minBucket = -1; float minCost = INF_F;
        for (int i = 0; i < NS; ++i)
            if (costs[i] < minCost) { minCost = costs[i]; minBucket = i; }
        float leafCost = (float) n;
        minCost = 0.5f + minCost / bounds.surfaceArea();
        if ((int) n > maxPrimsInNode || minCost < leafCost) {                                               // bvh.cpp:105-112
            Tri *m = std::partition(prims, prims + n, [=](const Tri &p) {
                int b = (int) (NB * centroidBounds.offset(p.centroid())[dim]);
                if (b == NB) b = NB - 1;
                return b <= minBucket;
            });
            mid = m - prims;
        } else {
            return makeLeaf(prims, n, bounds, orderedOffset, ordered, node);
        }
    }
    node->children[0] = buildTree(prims, mid, totalNodes, orderedOffset, ordered, maxPrimsInNode);
    node->children[1] = buildTree(prims + mid, n - mid, totalNodes, orderedOffset, ordered, maxPrimsInNode);
    node->bbox = Box();
    node->bbox.pmin = vmin(node->children[0]->bbox.pmin, node->children[1]->bbox.pmin);   // AABB(a,b) aabb.hpp:28-31
    node->bbox.pmax = vmax(node->children[0]->bbox.pmax, node->children[1]->bbox.pmax);
    node->splitAxis = dim;
    node->numPrimitives = 0;
    return node;
}

int flatten(const BuildNode *node, std::vector<ora_bvh_node> &nodes, int *offset, int depth, int *maxDepth) {
    if (depth > *maxDepth) *maxDepth = depth;
    int my = (*offset)++;
    ora_bvh_node &ln = nodes[my];
    std::memset(&ln, 0, sizeof ln);
    ln.pmin[0] = node->bbox.pmin.x; ln.pmin[1] = node->bbox.pmin.y; ln.pmin[2] = node->bbox.pmin.z;
    ln.pmax[0] = node->bbox.pmax.x; ln.pmax[1] = node->bbox.pmax.y; ln.pmax[2] = node->bbox.pmax.z;
    if (node->numPrimitives > 0) {
        ln.offset = node->firstPrimOffset;
        ln.num_prims = (uint16_t) node->numPrimitives;
    } else {
        ln.axis = (uint8_t) node->splitAxis;
        ln.num_prims = 0;
        flatten(node->children[0], nodes, offset, depth + 1, maxDepth);
        int second = flatten(node->children[1], nodes, offset, depth + 1, maxDepth);
        nodes[my].offset = second;
    }
    return my;
}

void destroyTree(BuildNode *n) {
    if (!n) return;
    destroyTree(n->children[0]); destroyTree(n->children[1]);
    delete n;
}

// Scene::buildBVH scene.cpp:96-135
void buildBVH(ora_scene &s, int maxPrimsInNode) {
    size_t n = s.triangles.size();
    s.ordered.assign(n, Tri());
    std::vector<Tri> work(n);
    for (size_t i = 0; i < n; ++i) {
        work[i].index = s.triangles[i].index;
        work[i].meshIndex = s.triangles[i].meshIndex;
        work[i].bounds = s.meshes[work[i].meshIndex].tBounds(work[i].index);
    }
    int totalNodes = 0, orderedOffset = 0;
    BuildNode *root = n ? buildTree(work.data(), n, &totalNodes, &orderedOffset, s.ordered, maxPrimsInNode) : nullptr;
    s.nodes.assign(totalNodes, ora_bvh_node());
    int offset = 0; s.maxDepth = 0;
    if (root) flatten(root, s.nodes, &offset, 0, &s.maxDepth);
    destroyTree(root);
    // getSceneRadius = |diagonal of root box| / 2 (scene.hpp:81-84); patch DISTANT lights (scene.cpp:128-134)
    if (!s.nodes.empty()) {
        V3 d = v3(s.nodes[0].pmax[0], s.nodes[0].pmax[1], s.nodes[0].pmax[2]) -
               v3(s.nodes[0].pmin[0], s.nodes[0].pmin[1], s.nodes[0].pmin[2]);
        s.radius = len(d) / 2;
    }
    for (auto &l : s.lights)
        if (l.type == 1) l.scene_radius = s.radius;
}

// ---------------------------------------------------------------------------------------------
// Moller-Trumbore (mesh.hpp:106-192)
// ---------------------------------------------------------------------------------------------
// returns true and b1,b2,root when the triangle is hit with tmin < root < tmax (Interval::surrounds, strict)
inline bool triTest(const Mesh &m, int index, V3 o, V3 d, float tmin, float tmax, float &b1, float &b2, float &root) {
    V3 v0, v1, v2;
    m.getVertices(index, v0, v1, v2);
    V3 v0v1 = v1 - v0;
    V3 v0v2 = v2 - v0;
    V3 pvec = cross(d, v0v2);
    float det = dot(v0v1, pvec);
    if (std::fabs((double) det) < 1e-8) return false;            // double literal compare, mesh.hpp:114
    float invDet = 1 / det;
    V3 tvec = o - v0;
    b1 = dot(tvec, pvec) * invDet;
    if (b1 < 0 || b1 > 1) return false;
    V3 qvec = cross(tvec, v0v1);
    b2 = dot(d, qvec) * invDet;
    if (b2 < 0 || b1 + b2 > 1) return false;
    root = dot(v0v2, qvec) * invDet;
    if (!(tmin < root && root < tmax)) return false;
    return true;
}

inline void fillRecord(const ora_scene &s, const Tri &tri, int prim, V3 o, V3 d, float root, float b1, float b2, Hit &rec) {
    const Mesh &m = s.meshes[tri.meshIndex];
    rec.t = root;
    rec.point = o + root * d;                                     // r.at(root) mesh.hpp:130
    rec.material = m.material;
    V3 n0, n1, n2; m.getNormals(tri.index, n0, n1, n2);
    float b0 = (1 - b1 - b2);
    V3 n = b0 * n0 + b1 * n1 + b2 * n2;                           // not renormalised
    rec.frontFace = dot(d, n) < 0;                                // material.hpp:36-39
    rec.normal = rec.frontFace ? n : -n;
    V2 a, b, c; m.getUVs(tri.index, a, b, c);
    rec.uv = {a.x * b0 + b.x * b1 + c.x * b2, a.y * b0 + b.y * b1 + c.y * b2};  // mesh.hpp:145
    rec.prim = prim; rec.b1 = b1; rec.b2 = b2;
}

// Scene::closestHit scene.cpp:10-55
bool closestHit(const ora_scene &s, V3 o, V3 d, float tmin, float tmax, Hit &rec, Counters *cnt) {
    if (s.nodes.empty()) return false;
    V3 invDir = 1 / d;
    int dirIsNeg[3] = {invDir.x < 0, invDir.y < 0, invDir.z < 0};
    int toVisit = 0, cur = 0, stack[64];
    bool hitAnything = false;
    if (cnt) cnt->n_closest++;
    while (true) {
        const ora_bvh_node &node = s.nodes[cur];
        if (cnt) cnt->n_nodes_closest++;
        if (g_node_hist) g_node_hist[cur].fetch_add(1, std::memory_order_relaxed);
        if (aabbHit(node.pmin, node.pmax, o, d, tmin, tmax)) {
            if (node.num_prims > 0) {
                for (int i = 0; i < node.num_prims; ++i) {
                    int prim = node.offset + i;
                    const Tri &tri = s.ordered[prim];
                    float b1, b2, root;
                    if (cnt) cnt->n_tri_closest++;
                    if (triTest(s.meshes[tri.meshIndex], tri.index, o, d, tmin, tmax, b1, b2, root)) {
                        fillRecord(s, tri, prim, o, d, root, b1, b2, rec);
                        hitAnything = true;
                        tmax = rec.t;
                        if (cnt) cnt->n_accept++;
                    }
                }
                if (toVisit == 0) break;
                cur = stack[--toVisit];
            } else {
                if (dirIsNeg[node.axis]) { stack[toVisit++] = cur + 1; cur = node.offset; }
                else                     { stack[toVisit++] = node.offset; cur = cur + 1; }
            }
        } else {
            if (toVisit == 0) break;
            cur = stack[--toVisit];
        }
    }
    return hitAnything;
}

// Scene::anyHit scene.cpp:57-94
bool anyHit(const ora_scene &s, V3 o, V3 d, float tmin, float tmax, Counters *cnt) {
    if (s.nodes.empty()) return false;
    V3 invDir = 1 / d;
    int dirIsNeg[3] = {invDir.x < 0, invDir.y < 0, invDir.z < 0};
    int toVisit = 0, cur = 0, stack[64];
    if (cnt) cnt->n_any++;
    while (true) {
        const ora_bvh_node &node = s.nodes[cur];
        if (cnt) cnt->n_nodes_any++;
        if (g_node_hist) g_node_hist[cur].fetch_add(1, std::memory_order_relaxed);
        if (aabbHit(node.pmin, node.pmax, o, d, tmin, tmax)) {
            if (node.num_prims > 0) {
                for (int i = 0; i < node.num_prims; ++i) {
                    const Tri &tri = s.ordered[node.offset + i];
                    float b1, b2, root;
                    if (cnt) cnt->n_tri_any++;
                    if (triTest(s.meshes[tri.meshIndex], tri.index, o, d, tmin, tmax, b1, b2, root)) return true;
                }
                if (toVisit == 0) break;
                cur = stack[--toVisit];
            } else {
                if (dirIsNeg[node.axis]) { stack[toVisit++] = cur + 1; cur = node.offset; }
                else                     { stack[toVisit++] = node.offset; cur = cur + 1; }
            }
        } else {
            if (toVisit == 0) break;
            cur = stack[--toVisit];
        }
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// Textures (image.hpp:140-160) and sRGB decode (util/color.hpp:13-23)
// ---------------------------------------------------------------------------------------------
inline V3 getTexel(const Texture &t, V2 uv) {
    int x = (int) (uv.x * t.w);
    int y = (int) (uv.y * t.h);
    int wu = x % t.w; if (wu < 0) wu += t.w;
    int wv = y % t.h; if (wv < 0) wv += t.h;
    const float *p = &t.data[(size_t) (wv * t.w + wu) * t.c];
    return {p[0], p[1], p[2]};
}
inline V3 sRGBToLinear(V3 c) {
    float in[3] = {c.x, c.y, c.z}, out[3];
    for (int i = 0; i < 3; ++i)
        out[i] = in[i] <= 0.04045f ? in[i] / 12.92f : std::pow((in[i] + 0.055f) / 1.055f, 2.4f);
    return {out[0], out[1], out[2]};
}

// ---------------------------------------------------------------------------------------------
// Frame::fromZ (Duff et al. branchless ONB)
// ---------------------------------------------------------------------------------------------
struct Frame {
    V3 x, y, z;
    static Frame fromZ(V3 n) {
        float sign = std::copysign(1.0f, n.z);
        float a = -1 / (sign + n.z);
        float b = n.x * n.y * a;
        Frame f;
        f.x = {1 + sign * sqr(n.x) * a, sign * b, -sign * n.x};
        f.y = {b, sign + sqr(n.y) * a, -n.y};
        f.z = n;
        return f;
    }
    V3 toLocal(V3 v) const { return {dot(v, x), dot(v, y), dot(v, z)}; }
    V3 toWorld(V3 v) const { return v.x * x + v.y * y + v.z * z; }
};

// ---------------------------------------------------------------------------------------------
// Fresnel / helpers (bsdf/bxdf.hpp:7-116), Complex (util/complex.hpp)
// ---------------------------------------------------------------------------------------------
inline V3 reflect(V3 wo, V3 n) { return -wo + 2 * dot(wo, n) * n; }            // bxdf.hpp:7-9

inline bool refract(V3 wi, V3 n, float eta, float *etap, V3 &wt) {             // bxdf.hpp:20-41
    float cosTheta_i = dot(wi, n);
    if (cosTheta_i < 0) { eta = 1 / eta; cosTheta_i = -cosTheta_i; n = -n; }
    if (etap) *etap = eta;
    float radicand = fmax2(0.0f, 1 - sqr(cosTheta_i)) / (eta * eta);
    if (radicand >= 1) return false;
    float cosTheta_t = safeSqrt(1 - radicand);
    wt = -wi / eta + (cosTheta_i / eta - cosTheta_t) * n;
    return true;
}
inline V3 schlick(V3 wo, V3 wm, V3 R) {                                        // bxdf.hpp:43-48
    float c = absdot(wo, wm);
    float m = 1 - c;
    float m2 = m * m;
    return R + (1.0f - R) * m2 * m2 * m;
}
inline float fresnelDielectric(float cosTheta_i, float eta) {                 // bxdf.hpp:56-77
    cosTheta_i = clampf(cosTheta_i, -1, 1);
    if (cosTheta_i < 0) { eta = 1 / eta; cosTheta_i = -cosTheta_i; }
    float radicand = (1 - cosTheta_i * cosTheta_i) / (eta * eta);
    if (radicand >= 1) return 1.0f;
    float cosTheta_t = safeSqrt(1 - radicand);
    float r_par  = (eta * cosTheta_i - cosTheta_t) / (eta * cosTheta_i + cosTheta_t);
    float r_perp = (cosTheta_i - eta * cosTheta_t) / (cosTheta_i + eta * cosTheta_t);
    return (r_par * r_par + r_perp * r_perp) / 2;
}
struct Cx { float r, i; };
inline Cx cx(float r, float i) { return {r, i}; }
inline Cx cmul(Cx a, Cx b) { return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }             // complex.hpp:26-28
inline Cx cdiv(Cx a, Cx c) {                                                                       // complex.hpp:30-33
    float scale = 1 / (c.r * c.r + c.i * c.i);
    return {(a.r * c.r + a.i * c.i) * scale, (a.i * c.r - a.r * c.i) * scale};
}
inline Cx cadd(Cx a, Cx b) { return {a.r + b.r, a.i + b.i}; }
inline Cx csub(Cx a, Cx b) { return {a.r - b.r, a.i - b.i}; }
inline float cnorm(Cx c) { return c.r * c.r + c.i * c.i; }                                         // complex.hpp:41-43
inline Cx csqrt(Cx c) {                                                                            // complex.hpp:49-57
    float n = std::sqrt(cnorm(c));
    float t1 = std::sqrt(0.5f * (n + std::fabs(c.r)));
    float t2 = 0.5f * c.i / t1;
    if (n == 0) return {0, 0};
    if (c.r >= 0) return {t1, t2};
    return {std::fabs(t2), std::copysign(t1, c.i)};
}
inline float fresnelComplex(float cosTheta_i, Cx eta) {                                           // bxdf.hpp:85-100
    cosTheta_i = clampf(cosTheta_i, 0, 1);
    float numerator = 1 - cosTheta_i * cosTheta_i;
    Cx radicand = cdiv(cx(numerator, 0), cmul(eta, eta));          // float / Complex = Complex(f)/c
    Cx one_minus = {1 - radicand.r, -radicand.i};                  // float - Complex
    Cx cosTheta_t = csqrt(one_minus);
    Cx eci = cmul(eta, cx(cosTheta_i, 0));                         // Complex * float = *this * Complex(f)
    Cx r_par = cdiv(csub(eci, cosTheta_t), cadd(eci, cosTheta_t));
    Cx ect = cmul(eta, cosTheta_t);
    Cx r_perp = cdiv(cx(cosTheta_i - ect.r, -ect.i), cx(cosTheta_i + ect.r, ect.i));
    return (cnorm(r_par) + cnorm(r_perp)) / 2;
}
inline V3 fresnelComplexRGB(float c, V3 eta, V3 k) {                                               // bxdf.hpp:110-116
    return {fresnelComplex(c, cx(eta.x, k.x)), fresnelComplex(c, cx(eta.y, k.y)), fresnelComplex(c, cx(eta.z, k.z))};
}

// ---------------------------------------------------------------------------------------------
// GGX (bsdf/microfacet.hpp)
// ---------------------------------------------------------------------------------------------
struct GGX {
    float ax, ay;
    bool smooth() const { return fmax2(ax, ay) < 1e-3f; }                                          // microfacet.hpp:23-25
    float D(V3 wm) const {                                                                         // microfacet.hpp:32-41
        float t2 = tan2Theta(wm);
        if (std::isinf(t2)) return 0;
        float cos4 = sqr(cos2Theta(wm));
        if (cos4 < 1e-6f) return 0;
        float e = t2 * (sqr(cosPhi(wm) / ax) + sqr(sinPhi(wm) / ay));
        return 1 / (PI_F * ax * ay * cos4 * sqr(1 + e));
    }
    float lambda(V3 w) const {                                                                     // microfacet.hpp:62-67
        float t2 = tan2Theta(w);
        if (std::isinf(t2)) return 0;
        float alpha2 = sqr(ax * cosPhi(w)) + sqr(ay * sinPhi(w));
        return 0.5f * (std::sqrt(1 + alpha2 * t2) - 1);
    }
    float G1(V3 w) const { return 1 / (1 + lambda(w)); }                                           // microfacet.hpp:74
    float G(V3 wo, V3 wi) const { return 1 / (1 + lambda(wo) + lambda(wi)); }                      // microfacet.hpp:82-84
    float pdf(V3 w, V3 wm) const { return G1(w) / absCosTheta(w) * D(wm) * absdot(w, wm); }        // microfacet.hpp:49-55
    V3 sampleWm(V3 w, V2 u) const {                                                                // microfacet.hpp:86-106
        V3 wh = normalize(v3(ax * w.x, ay * w.y, w.z));
        if (wh.z < 0) wh = -wh;
        V3 t1 = (wh.z < 0.99999f) ? normalize(cross(v3(0, 0, 1), wh)) : v3(1, 0, 0);
        V3 t2 = cross(wh, t1);
        V2 p = sampleUniformDiskPolar(u);
        float h = std::sqrt(1 - sqr(p.x));
        p.y = lerpf(h, p.y, (1 + wh.z) / 2);
        float pz = std::sqrt(fmax2(0.0f, 1.0f - (p.x * p.x + p.y * p.y)));
        V3 nh = p.x * t1 + p.y * t2 + pz * wh;
        return normalize(v3(ax * nh.x, ay * nh.y, fmax2(1e-6f, nh.z)));
    }
};

struct BSample { V3 f; V3 wi; float pdf; bool specular = false; };   // specular = BSDFSample::isSpecular (bxdf.hpp:123)

// DiffuseBxDF (bsdf/diffuse.hpp)
inline V3 diffuseEval(V3 R, V3 wo, V3 wi) { return sameHemisphere(wo, wi) ? R * INV_PI : v3(0); }
inline bool diffuseSample(V3 R, V3 wo, V2 u, BSample &s) {
    V3 wi = sampleCosineHemisphere(u);
    if (wo.z < 0) wi.z *= -1;
    s = {R * INV_PI, wi, cosineHemispherePDF(absCosTheta(wi))};
    return true;
}
inline float diffusePdf(V3 wo, V3 wi) { return sameHemisphere(wo, wi) ? cosineHemispherePDF(absCosTheta(wi)) : 0; }

// ConductorBxDF (bsdf/conductor.hpp)
inline V3 conductorEval(GGX mf, V3 eta, V3 k, V3 wo, V3 wi) {                                      // conductor.hpp:13-29
    if (mf.smooth()) return v3(0);
    float co = absCosTheta(wo), ci = absCosTheta(wi);
    if (co == 0 || ci == 0) return v3(0);
    V3 wm = wi + wo;
    if (lenSqr(wm) == 0) return v3(0);
    wm = normalize(wm);
    V3 F = fresnelComplexRGB(absdot(wo, wm), eta, k);
    return mf.D(wm) * F * mf.G(wo, wi) / (4 * ci * co);
}
inline bool conductorSample(GGX mf, V3 eta, V3 k, V3 wo, V2 u, BSample &s) {                       // conductor.hpp:31-59
    if (mf.smooth()) {
        V3 wi = v3(-wo.x, -wo.y, wo.z);
        float ci = absCosTheta(wi);
        V3 f = fresnelComplexRGB(ci, eta, k) / ci;
        s = {f, wi, 1};
        return true;
    }
    if (wo.z == 0) return false;
    V3 wm = mf.sampleWm(wo, u);
    V3 wi = reflect(wo, wm);
    if (!sameHemisphere(wo, wi)) return false;
    float co = absCosTheta(wo), ci = absCosTheta(wi);
    if (co == 0 || ci == 0) return false;
    float pdf = mf.pdf(wo, wm) / (4 * absdot(wo, wm));
    V3 F = fresnelComplexRGB(absdot(wo, wm), eta, k);
    V3 f = mf.D(wm) * F * mf.G(wo, wi) / (4 * ci * co);
    s = {f, wi, pdf};
    return true;
}
inline float conductorPdf(GGX mf, V3 wo, V3 wi) {                                                  // conductor.hpp:61-68
    if (mf.smooth()) return 0;
    if (!sameHemisphere(wo, wi)) return 0;
    V3 wm = wo + wi;
    if (lenSqr(wm) == 0) return 0;
    wm = faceForward(normalize(wm), v3(0, 0, 1));
    return mf.pdf(wo, wm) / (4 * absdot(wo, wm));
}

// DielectricBxDF (bsdf/dielectric.hpp:5-161)
inline V3 dielectricEval(GGX mf, float eta, V3 wo, V3 wi) {                                        // dielectric.hpp:9-40
    if (eta == 1 || mf.smooth()) return v3(0);
    float co = absCosTheta(wo), ci = absCosTheta(wi);   // NB: abs => "reflect" is always true unless a cos is 0
    bool refl = co * ci > 0;
    float etap = 1;
    if (!refl) etap = co > 0 ? eta : 1 / eta;
    V3 wm = wi * etap + wo;
    if (ci == 0 || co == 0 || lenSqr(wm) == 0) return v3(0);
    wm = faceForward(normalize(wm), v3(0, 0, 1));
    if (dot(wm, wi) * ci < 0 || dot(wm, wo) * co < 0) return v3(0);
    float F = fresnelDielectric(dot(wo, wm), eta);
    if (refl) {
        return v3(mf.D(wm) * F * mf.G(wo, wi) / std::fabs(4 * ci * co));
    } else {
        float a = mf.D(wm) * (1 - F) * mf.G(wo, wi) * std::fabs(dot(wi, wm) * dot(wo, wm));
        float b = sqr(dot(wi, wm) + dot(wo, wm) / etap) * std::fabs(ci * co);
        return v3(a / b);
    }
}
inline bool dielectricSample(GGX mf, float eta, V3 wo, float uc, V2 u, BSample &s) {               // dielectric.hpp:42-108
    const bool isSpecular = mf.smooth();                                        // dielectric.hpp:43: s = {f, w_i, pdf, 0, isSpecular} in all four returns
    if (eta == 1 || mf.smooth()) {
        float R = fresnelDielectric(cosTheta(wo), eta);
        float T = 1 - R;
        float p = R / (R + T);
        if (uc < p) {
            V3 wi = v3(-wo.x, -wo.y, wo.z);
            s = {v3(R / absCosTheta(wi)), wi, p, isSpecular};
            return true;
        } else {
            V3 wi; float etap;
            if (!refract(wo, v3(0, 0, 1), eta, &etap, wi)) return false;
            s = {v3(T / absCosTheta(wi)), wi, 1 - p, isSpecular};
            return true;
        }
    }
    V3 wm = mf.sampleWm(wo, u);
    float R = fresnelDielectric(dot(wo, wm), eta);
    float T = 1 - R;
    float p = R / (R + T);
    if (uc < p) {
        V3 wi = reflect(wo, wm);
        if (!sameHemisphere(wo, wi)) return false;
        float pdf = mf.pdf(wo, wm) / (4 * absdot(wo, wm)) * p;
        float f = mf.D(wm) * mf.G(wo, wi) * R / (4 * absCosTheta(wi) * absCosTheta(wo));
        s = {v3(f), wi, pdf, isSpecular};
        return true;
    } else {
        float etap; V3 wi = v3(0);     // w_i default-constructed (0,0,0) when refract fails (dielectric.hpp:93-95)
        bool tir = !refract(wo, wm, eta, &etap, wi);
        if (sameHemisphere(wo, wi) || wi.z == 0 || tir) return false;
        float dn = absdot(wi, wm) / sqr(dot(wi, wm) + dot(wo, wm) / etap);
        float pdf = mf.pdf(wo, wm) * dn * (1 - p);
        float f = mf.D(wm) * T * mf.G(wo, wi) * std::fabs(dot(wi, wm) * dot(wo, wm));
        f /= sqr(dot(wi, wm) + dot(wm, wo) / etap) * std::fabs(cosTheta(wi) * cosTheta(wo));
        s = {v3(f), wi, pdf, isSpecular};
        return true;
    }
}
inline float dielectricPdf(GGX mf, float eta, V3 wo, V3 wi) {                                      // dielectric.hpp:110-157
    if (eta == 1 || mf.smooth()) return 0;
    float co = absCosTheta(wo), ci = absCosTheta(wi);
    bool refl = co * ci > 0;
    float etap = 1;
    if (!refl) etap = co > 0 ? eta : 1 / eta;
    V3 wm = wi * etap + wo;
    if (ci == 0 || co == 0 || lenSqr(wm) == 0) return 0;
    wm = faceForward(normalize(wm), v3(0, 0, 1));
    if (dot(wm, wi) * ci < 0 || dot(wm, wo) * co < 0) return 0;
    float R = fresnelDielectric(dot(wo, wm), eta);
    float T = 1 - R;
    if (refl) return mf.pdf(wo, wm) / (4 * absdot(wo, wm)) * (R / (R + T));
    float dn = absdot(wi, wm) / sqr(dot(wi, wm) + dot(wo, wm) / etap);
    return mf.pdf(wo, wm) * dn * (T / (R + T));
}

// MetallicRoughnessBxDF (bsdf/gltf.hpp:9-123); mf = GGX(roughness2, roughness2)
struct MR { GGX mf; V3 albedo; float metallic; };
inline V3 mrEval(const MR &b, V3 wo, V3 wi) {                                                      // gltf.hpp:16-34
    float co = absCosTheta(wo), ci = absCosTheta(wi);
    if (co == 0 || ci == 0) return v3(0);
    V3 cDiff = lerp3(b.albedo, v3(0), b.metallic);
    V3 f0 = lerp3(v3(0.04f), b.albedo, b.metallic);
    V3 wm = wi + wo;
    if (lenSqr(wm) == 0) return v3(0);
    wm = normalize(wm);
    V3 F = schlick(wo, wm, f0);
    V3 fDiffuse = (1 - F) * cDiff * INV_PI;
    V3 fSpecular = b.mf.D(wm) * F * b.mf.G(wo, wi) / (4 * absCosTheta(wi) * absCosTheta(wo));
    return fDiffuse + fSpecular;
}
inline float mrSpecProb(const MR &b, V3 wo) {                                                      // gltf.hpp:42-50 / 99-107
    V3 f0 = lerp3(v3(0.04f), b.albedo, b.metallic);
    V3 F = schlick(wo, v3(0, 0, 1), f0);
    float specularWeight = (F.x + F.y + F.z) / 3;          // Vec3::average
    float diffuseWeight = (1 - b.metallic) * (1 - specularWeight);
    float total = specularWeight + diffuseWeight;
    float p = 1.0f;
    if (total > 0) p = specularWeight / total;
    return p;
}
inline bool mrSample(const MR &b, V3 wo, float uc, V2 u, BSample &s) {                             // gltf.hpp:36-90
    float co = absCosTheta(wo);
    if (co == 0) return false;
    V3 cDiff = lerp3(b.albedo, v3(0), b.metallic);
    V3 f0 = lerp3(v3(0.04f), b.albedo, b.metallic);
    float p = mrSpecProb(b, wo);
    V3 wi, wm; float pdf;
    if (uc < p) {
        if (wo.z == 0) return false;
        wm = b.mf.sampleWm(wo, u);
        wi = -wo + 2 * dot(wo, wm) * wm;                     // jtx::reflect
        if (!sameHemisphere(wo, wi)) return false;
        float ci = absCosTheta(wi);
        if (ci == 0) return false;
        pdf = b.mf.pdf(wo, wm) / (4 * absdot(wo, wm));
    } else {
        wi = sampleCosineHemisphere(u);
        if (wo.z < 0) wi.z *= -1;
        wm = wi + wo;
        if (lenSqr(wm) == 0) return false;
        wm = normalize(wm);
        pdf = cosineHemispherePDF(absCosTheta(wi));
    }
    V3 F = schlick(wo, wm, f0);
    V3 fDiffuse = (1 - F) * (cDiff / PI_F);
    V3 fSpecular = b.mf.D(wm) * F * b.mf.G(wo, wi) / (4 * absCosTheta(wi) * absCosTheta(wo));
    s.pdf = pdf; s.wi = wi; s.f = fDiffuse + fSpecular;
    return true;
}
inline float mrPdf(const MR &b, V3 wo, V3 wi) {                                                    // gltf.hpp:92-117
    if (!sameHemisphere(wo, wi)) return 0;
    float co = absCosTheta(wo);
    if (co == 0) return 0;
    float p = mrSpecProb(b, wo);
    V3 wm = wi + wo;
    if (lenSqr(wm) == 0) return 0.0f;
    wm = faceForward(normalize(wm), v3(0, 0, 1));
    float specularPdf = b.mf.pdf(wo, wm) / (4 * absdot(wo, wm));
    float diffusePdf = cosineHemispherePDF(absCosTheta(wi));
    return p * specularPdf + (1 - p) * diffusePdf;
}

// ThinDielectricBxDF::sample (dielectric.hpp:173-199); evaluate = {}, pdf = 0 (dielectric.hpp:167-169, 201-203).
// The reference never dispatches it (bxdf.cpp has no case): Material type 4 is this build's way of reaching it.
inline bool thinDielectricSample(float eta, V3 wo, float uc, BSample &s) {
    float R = fresnelDielectric(wo.z, eta);
    float T = 1 - R;
    if (R < 1) {
        R += (T * T * R) / (1 - R * R);
        T = 1 - R;
    }
    const float p = R / (R + T);
    if (uc < p) {
        V3 wi = v3(-wo.x, -wo.y, wo.z);
        s.f = v3(R / absCosTheta(wi)); s.wi = wi; s.pdf = p;
        return true;
    }
    V3 wi = -wo;
    s.f = v3(T / absCosTheta(wi)); s.wi = wi; s.pdf = 1 - p;
    return true;
}

// ---------------------------------------------------------------------------------------------
// sampleBxdf / evalBxdf / pdfBxdf (bsdf/bxdf.cpp:9-166)
// ---------------------------------------------------------------------------------------------
inline V3 a3(const float *p) { return {p[0], p[1], p[2]}; }

inline V3 albedoOf(const ora_scene &s, const ora_material &m, V2 uv) {
    V3 albedo = a3(m.albedo);
    if (m.albedo_tex != -1) albedo = sRGBToLinear(getTexel(s.textures[m.albedo_tex], uv));
    return albedo;
}
inline void mrParams(const ora_scene &s, const ora_material &m, V2 uv, float &metallic, float &roughness) {
    metallic = m.alpha_x; roughness = m.alpha_y;
    if (m.mr_tex != -1) { V3 mr = getTexel(s.textures[m.mr_tex], uv); roughness = mr.y; metallic = mr.z; }
}

// the code path a material's sampleBxdf / evalBxdf / pdfBxdf calls take (ora_counters::n_shade_class): dielectric.hpp:44, conductor.hpp:33
inline int bxdfClass(const ora_material &m) {
    const bool smooth = GGX{m.alpha_x, m.alpha_y}.smooth();
    if (m.type == 1) return (m.ior[0] == 1.0f || smooth) ? 5 : 1;
    if (m.type == 2) return smooth ? 6 : 2;
    return m.type;
}

bool sampleBxdf(const ora_scene &s, const ora_material &m, V3 normal, V2 uv, V3 wo, float uc, V2 u, BSample &out) {
    Frame fr = Frame::fromZ(normal);
    V3 wol = fr.toLocal(wo);
    if (wol.z == 0) return false;
    bool ok = false;
    out.specular = false;                                                       // BSDFSample s; / s = {f, w_i, pdf}: default false
    if (m.type == 3) {
        float metallic, roughness; mrParams(s, m, uv, metallic, roughness);
        MR b{GGX{roughness * roughness, roughness * roughness}, albedoOf(s, m, uv), metallic};
        ok = mrSample(b, wol, uc, u, out);
    } else if (m.type == 0) {
        ok = diffuseSample(albedoOf(s, m, uv), wol, u, out);
    } else if (m.type == 2) {
        ok = conductorSample(GGX{m.alpha_x, m.alpha_y}, a3(m.ior), a3(m.k), wol, u, out);
    } else if (m.type == 1) {
        ok = dielectricSample(GGX{m.alpha_x, m.alpha_y}, m.ior[0], wol, uc, u, out);
    } else if (m.type == 4) {
        ok = thinDielectricSample(m.ior[0], wol, uc, out);
    }
    if (!ok) return false;
    if (!nonzero(out.f) || out.pdf == 0 || out.wi.z == 0) return false;
    out.wi = fr.toWorld(out.wi);
    return true;
}
V3 evalBxdf(const ora_scene &s, const ora_material &m, V3 normal, V2 uv, V3 wo, V3 wi) {
    Frame fr = Frame::fromZ(normal);
    V3 wol = fr.toLocal(wo), wil = fr.toLocal(wi);
    if (wol.z == 0 || wil.z == 0) return v3(0);
    if (m.type == 3) {
        float metallic, roughness; mrParams(s, m, uv, metallic, roughness);
        MR b{GGX{roughness * roughness, roughness * roughness}, albedoOf(s, m, uv), metallic};
        return mrEval(b, wol, wil);
    }
    if (m.type == 0) return diffuseEval(albedoOf(s, m, uv), wol, wil);
    if (m.type == 2) return conductorEval(GGX{m.alpha_x, m.alpha_y}, a3(m.ior), a3(m.k), wol, wil);
    if (m.type == 1) return dielectricEval(GGX{m.alpha_x, m.alpha_y}, m.ior[0], wol, wil);
    return v3(0);
}
float pdfBxdf(const ora_scene &s, const ora_material &m, V3 normal, V2 uv, V3 wo, V3 wi) {
    Frame fr = Frame::fromZ(normal);
    V3 wol = fr.toLocal(wo), wil = fr.toLocal(wi);
    if (wol.z == 0 || wil.z == 0) return 0;
    if (m.type == 3) {
        float metallic, roughness; mrParams(s, m, uv, metallic, roughness);
        MR b{GGX{roughness * roughness, roughness * roughness}, a3(m.albedo), metallic};   // constant albedo: bxdf.cpp:146
        return mrPdf(b, wol, wil);
    }
    if (m.type == 0) return diffusePdf(wol, wil);
    if (m.type == 2) return conductorPdf(GGX{m.alpha_x, m.alpha_y}, wol, wil);
    if (m.type == 1) return dielectricPdf(GGX{m.alpha_x, m.alpha_y}, m.ior[0], wol, wil);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Lights (lights/lights.hpp:36-54) and sampleLights (integrator.cpp:134-169)
// ---------------------------------------------------------------------------------------------
struct LightSample { V3 p, radiance, wi; float pdf; };

inline bool lightSample(const ora_light &l, V3 p, LightSample &ls) {
    V3 pos = a3(l.position), I = a3(l.intensity);
    if (l.type == 0) {
        ls.p = pos;
        ls.wi = normalize(pos - p);
        ls.radiance = l.scale * I / lenSqr(pos - p);              // distanceSqr(position, ctx.p)
        ls.pdf = 1;
        return true;
    }
    if (l.type == 1) {
        ls.p = p - pos * 2 * l.scene_radius;
        ls.wi = -pos;
        ls.radiance = l.scale * I;
        ls.pdf = 1;
        return true;
    }
    return false;
}

inline float powerHeuristic(float nf, float fPdf, float ng, float gPdf) {     // integrator.cpp:6-10
    float f = nf * fPdf, g = ng * gPdf;
    return f * f / (f * f + g * g);
}

V3 sampleLights(const ora_scene &s, V3 rayDir, const Hit &rec, Rng &rng, Counters *cnt) {
    int n = (int) s.lights.size();
    uint32_t idx = rng.sampleRange(n - 1);                                    // integrator.cpp:135 (Q2)
    const ora_light &light = s.lights[idx];
    (void) rng.f(); (void) rng.f();                                           // u, consumed (integrator.cpp:142)
    LightSample ls;
    if (lightSample(light, rec.point, ls)) {
        V3 sOrigin = rec.point + rec.normal * RAY_EPSILON;
        float lDist = len(rec.point - ls.p);                                  // distance(record.point, ls.p)
        bool occluded = anyHit(s, sOrigin, ls.wi, 0.0f, lDist - RAY_EPSILON, cnt);
        if (!occluded) {
            V3 wo = -rayDir, wi = ls.wi;
            const ora_material &m = s.materials[rec.material];
            if (cnt) cnt->eval(bxdfClass(m));
            V3 f = evalBxdf(s, m, rec.normal, rec.uv, wo, wi) * absdot(wi, rec.normal);
            float pb = pdfBxdf(s, m, rec.normal, rec.uv, wo, wi);
            float pl = 1.0f / (float) n * ls.pdf;
            float misWeight = powerHeuristic(1, pl, 1, pb);                   // always applied (Q10)
            return misWeight * f * ls.radiance / pl;
        }
    }
    return v3(0);
}

// integrateMIS integrator.cpp:171-216
V3 integrateMIS(const ora_scene &s, V3 o, V3 d, int maxDepth, Rng &rng, Counters *cnt) {
    V3 radiance = v3(0), beta = v3(1);
    int depth = 0;
    Hit rec;
    bool hasLights = !s.lights.empty();
    while (true) {
        bool hit = closestHit(s, o, d, 0.001f, INF_F, rec, cnt);
        if (!hit) { radiance = radiance + beta * s.sky; break; }
        if (depth++ == maxDepth) break;
        if (hasLights) radiance = radiance + beta * sampleLights(s, d, rec, rng, cnt);
        V3 wo = -d;
        float u = rng.f();
        V2 u2; u2.x = rng.f(); u2.y = rng.f();
        BSample bs;
        if (cnt) cnt->shade(bxdfClass(s.materials[rec.material]));
        if (!sampleBxdf(s, s.materials[rec.material], rec.normal, rec.uv, wo, u, u2, bs)) break;
        if (bs.pdf > 0.0f) beta = beta * (bs.f * absdot(bs.wi, rec.normal) / bs.pdf);
        o = rec.point + bs.wi * RAY_EPSILON;
        d = bs.wi;
    }
    return radiance;
}

// integrate integrator.cpp:54-132: next-event estimation without MIS; emission and sky only after a specular bounce
V3 integrateNEE(const ora_scene &s, V3 o, V3 d, int maxDepth, Rng &rng, Counters *cnt) {
    V3 radiance = v3(0), beta = v3(1);
    int depth = 0;
    bool specularBounce = true;
    Hit rec;
    const int n = (int) s.lights.size();
    while (nonzero(beta)) {
        bool hit = closestHit(s, o, d, 0.001f, INF_F, rec, cnt);
        if (!hit) {
            if (specularBounce) radiance = radiance + beta * s.sky;
            break;
        }
        const ora_material &m = s.materials[rec.material];
        if (specularBounce) radiance = radiance + beta * a3(m.emission);
        if (depth++ == maxDepth) break;
        V3 wo = -d;
        {
            uint32_t idx = rng.sampleRange(n - 1);                              // integrator.cpp:85 (needs a light: the callers refuse n == 0)
            const ora_light &light = s.lights[idx];
            (void) rng.f(); (void) rng.f();                                     // Vec2f u, integrator.cpp:95
            LightSample ls;
            bool lightSampled = lightSample(light, rec.point, ls);
            if (lightSampled && ls.pdf > 0) {
                V3 wi = ls.wi;
                if (cnt) cnt->eval(bxdfClass(m));
                V3 f = evalBxdf(s, m, rec.normal, rec.uv, wo, wi) * absdot(wi, rec.normal);
                V3 sOrigin = rec.point + rec.normal * RAY_EPSILON;
                float lDist = len(rec.point - ls.p);
                if (nonzero(f) && !anyHit(s, sOrigin, ls.wi, 0.0f, lDist - RAY_EPSILON, cnt))
                    radiance = radiance + beta * f * ls.radiance / (ls.pdf * (1.0f / (float) n));
            }
        }
        float u = rng.f();
        V2 u2; u2.x = rng.f(); u2.y = rng.f();
        BSample bs;
        if (cnt) cnt->shade(bxdfClass(m));
        if (!sampleBxdf(s, m, rec.normal, rec.uv, wo, u, u2, bs)) break;
        beta = beta * (bs.f * absdot(bs.wi, rec.normal) / bs.pdf);            // unguarded, integrator.cpp:125
        specularBounce = bs.specular;
        o = rec.point + bs.wi * RAY_EPSILON;
        d = bs.wi;
    }
    return radiance;
}

// integrateBasic integrator.cpp:12-52: no light sampling, emission at every hit
V3 integrateBasic(const ora_scene &s, V3 o, V3 d, int maxDepth, Rng &rng, Counters *cnt) {
    V3 radiance = v3(0), beta = v3(1);
    int depth = 0;
    Hit rec;
    while (nonzero(beta)) {
        bool hit = closestHit(s, o, d, 0.001f, INF_F, rec, cnt);
        if (!hit) { radiance = radiance + beta * s.sky; break; }
        const ora_material &m = s.materials[rec.material];
        radiance = radiance + beta * a3(m.emission);
        if (depth++ == maxDepth) break;
        V3 wo = -d;
        float u = rng.f();
        V2 u2; u2.x = rng.f(); u2.y = rng.f();
        BSample bs;
        if (cnt) cnt->shade(bxdfClass(m));
        if (!sampleBxdf(s, m, rec.normal, rec.uv, wo, u, u2, bs)) break;
        beta = beta * (bs.f * absdot(bs.wi, rec.normal) / bs.pdf);
        o = rec.point + bs.wi * RAY_EPSILON;
        d = bs.wi;
    }
    return radiance;
}

std::atomic<int> g_li{0};      // which Li the render entry points use: 0 integrateMIS, 1 integrate, 2 integrateBasic (camera.cpp:104-106)

// ---------------------------------------------------------------------------------------------
// Camera (camera.cpp:7-31, camera.hpp:107-139)
// ---------------------------------------------------------------------------------------------
struct Cam {
    V3 center, vp00, du, dv, defocus_u, defocus_v;
    float defocusAngle;
    int xs, ys;
};
Cam camInit(const ora_camera_desc &c) {
    Cam k;
    V3 center = a3(c.center), target = a3(c.target), up = a3(c.up);
    float aspect = (float) c.width / (float) c.height;
    float yf = c.yfov * PI_F / 180.0f;                                         // radians(), rt.hpp:42-44
    float h = std::tan(yf / 2);
    float viewportHeight = 2 * h * c.focus_distance;
    float viewportWidth = viewportHeight * aspect;
    V3 w = normalize(center - target);
    V3 u = normalize(cross(up, w));
    V3 v = cross(w, u);
    V3 viewportU = viewportWidth * u;
    V3 viewportV = viewportHeight * v;
    k.du = viewportU / (float) c.width;
    k.dv = viewportV / (float) c.height;
    V3 upperLeft = center - (c.focus_distance * w) - viewportU / 2 - viewportV / 2;
    k.vp00 = upperLeft + 0.5f * (k.du + k.dv);
    float da = (c.defocus_angle / 2) * PI_F / 180.0f;
    float defocusRadius = c.focus_distance * std::tan(da);
    k.defocus_u = defocusRadius * u;
    k.defocus_v = defocusRadius * v;
    k.center = center;
    k.defocusAngle = c.defocus_angle;
    k.xs = c.x_pixel_samples; k.ys = c.y_pixel_samples;
    return k;
}
// getRay(i=col, j=row, stratum, rng) camera.hpp:127-139 (called (col,row): camera.cpp:103)
inline void camRay(const Cam &k, uint32_t col, uint32_t row, uint32_t stratum, Rng &rng, V3 &o, V3 &d) {
    uint32_t x = stratum % k.xs, y = stratum / k.xs;
    float dx = rng.f(), dy = rng.f();
    V2 off = {((float) x + dx) / (float) k.xs, ((float) y + dy) / (float) k.ys};
    V3 sample = k.vp00 + ((float) col + off.x) * k.du + ((float) row + off.y) * k.dv;
    V3 origin = k.center;
    if (!(k.defocusAngle <= 0)) {
        // RNG::sampleUnitDisc rand.hpp:179-184, sample<float>(-1,1) = min + (max-min)*u, x drawn first
        V3 p;
        while (true) {
            float px = -1.0f + (1.0f - -1.0f) * rng.f();
            float py = -1.0f + (1.0f - -1.0f) * rng.f();
            p = v3(px, py, 0);
            if (lenSqr(p) < 1) break;
        }
        origin = k.center + (p.x * k.defocus_u) + (p.y * k.defocus_v);
    }
    (void) rng.f();                                                           // ray time, consumed
    o = origin; d = sample - origin;                                          // not normalised (Q5)
}

inline V3 tracePixelSample(const ora_scene &s, const Cam &k, int maxDepth, uint32_t row, uint32_t col, uint32_t sample,
                           Counters *cnt) {
    Rng rng(row, col, sample + 1);                                            // camera.cpp:101
    V3 o, d; camRay(k, col, row, sample, rng, o, d);
    if (cnt) cnt->n_camera++;
    const int li = g_li.load(std::memory_order_relaxed);
    V3 c = li == 1 ? integrateNEE(s, o, d, maxDepth, rng, cnt) : li == 2 ? integrateBasic(s, o, d, maxDepth, rng, cnt)
                   : integrateMIS(s, o, d, maxDepth, rng, cnt);
    if (c.x > 1.0f) c.x = 1.0f;                                               // camera.cpp:110-112
    if (c.y > 1.0f) c.y = 1.0f;
    if (c.z > 1.0f) c.z = 1.0f;
    return c;
}

inline uint8_t toByte(float v) {                                              // image.hpp:9-16,47-52
    float g = v > 0 ? std::sqrt(v) : 0.0f;
    float c = clampf(g, 0.0f, 0.999f);
    return (uint8_t) (int) (255.999f * c);
}

} // namespace

// =============================================================================================
// C interface
// =============================================================================================
extern "C" {

void ora_render(const ora_scene *s, const ora_camera_desc *cam, int threads, int sample_begin, int sample_end,
                int reference_barriers, float *acc_rgb, uint8_t *img_rgb, ora_counters *counters);

void ora_set_sincos_mode(int mode) { g_sincos_mode.store(mode); }
void ora_set_integrator(int li) { g_li.store(li); }

// diagnostic: renders the frame once more and returns how often every BVH node was visited (closest + any)
void ora_node_histogram(const ora_scene *s, const ora_camera_desc *cam, int threads, uint64_t *out) {
    std::vector<std::atomic<uint64_t>> h(s->nodes.size());
    for (auto &x : h) x.store(0);
    g_node_hist = h.data();
    std::vector<float> acc(3 * (size_t) cam->width * cam->height);
    ora_render(s, cam, threads, 0, cam->x_pixel_samples * cam->y_pixel_samples, 0, acc.data(), nullptr, nullptr);
    g_node_hist = nullptr;
    for (size_t i = 0; i < h.size(); ++i) out[i] = h[i].load();
}

uint32_t ora_fnv1a_3(uint32_t x, uint32_t y, uint32_t n) { return fnv1a_3(x, y, n); }

void ora_rng_stream(uint32_t x, uint32_t y, uint32_t n, int count, uint32_t *out_u32, float *out_f32) {
    Rng a(x, y, n), b(x, y, n);
    for (int i = 0; i < count; ++i) {
        if (out_u32) out_u32[i] = a.advance();
        if (out_f32) out_f32[i] = b.f();
    }
}
uint32_t ora_rng_sample_range(uint32_t x, uint32_t y, uint32_t n, int skip, int range_arg) {
    Rng a(x, y, n);
    for (int i = 0; i < skip; ++i) a.advance();
    return a.sampleRange(range_arg);
}

void ora_sincos_batch(const float *x, int n, float *out_sin, float *out_cos) {
    for (int i = 0; i < n; ++i) det_sincos(x[i], &out_sin[i], &out_cos[i]);
}

ora_scene *ora_scene_create(const ora_scene_desc *d) {
    ora_scene *s = new ora_scene();
    s->sky = a3(d->sky_color);
    for (int i = 0; i < d->num_meshes; ++i) {
        const ora_mesh &m = d->meshes[i];
        Mesh mm;
        mm.numTris = m.num_triangles; mm.numVerts = m.num_vertices;
        mm.indices.assign(m.indices, m.indices + 3 * (size_t) m.num_triangles);
        mm.vertices.resize(m.num_vertices); mm.normals.resize(m.num_vertices);
        for (int v = 0; v < m.num_vertices; ++v) {
            mm.vertices[v] = a3(m.vertices + 3 * v);
            mm.normals[v] = a3(m.normals + 3 * v);
        }
        mm.hasUVs = m.uvs != nullptr;
        if (mm.hasUVs) {
            mm.uvs.resize(m.num_vertices);
            for (int v = 0; v < m.num_vertices; ++v) mm.uvs[v] = V2{m.uvs[2 * v], m.uvs[2 * v + 1]};
        }
        mm.material = m.material;
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) mm.m[r][c] = m.transform[4 * r + c];
        s->meshes.push_back(std::move(mm));
    }
    for (int i = 0; i < d->num_tri_refs; ++i) {
        Tri t; t.index = d->tri_refs[i].index; t.meshIndex = d->tri_refs[i].mesh_index;
        s->triangles.push_back(t);
    }
    s->materials.assign(d->materials, d->materials + d->num_materials);
    s->lights.assign(d->lights, d->lights + d->num_lights);
    for (int i = 0; i < d->num_textures; ++i) {
        Texture t; t.w = d->textures[i].width; t.h = d->textures[i].height; t.c = d->textures[i].channels;
        t.data.assign(d->textures[i].texels, d->textures[i].texels + (size_t) t.w * t.h * t.c);
        s->textures.push_back(std::move(t));
    }
    buildBVH(*s, d->max_prims_in_node > 0 ? d->max_prims_in_node : 1);
    return s;
}
void ora_scene_destroy(ora_scene *s) { delete s; }
int ora_scene_num_nodes(const ora_scene *s) { return (int) s->nodes.size(); }
int ora_scene_num_prims(const ora_scene *s) { return (int) s->ordered.size(); }
int ora_scene_max_depth(const ora_scene *s) { return s->maxDepth; }
float ora_scene_radius(const ora_scene *s) { return s->radius; }
void ora_scene_get_bvh(const ora_scene *s, ora_bvh_node *nodes_out, ora_tri_ref *refs_out) {
    if (nodes_out) std::memcpy(nodes_out, s->nodes.data(), s->nodes.size() * sizeof(ora_bvh_node));
    if (refs_out)
        for (size_t i = 0; i < s->ordered.size(); ++i) refs_out[i] = ora_tri_ref{s->ordered[i].index, s->ordered[i].meshIndex};
}

int ora_aabb_hit(const float pmin[3], const float pmax[3], const float o[3], const float d[3], float t0, float t1) {
    return aabbHit(pmin, pmax, a3(o), a3(d), t0, t1) ? 1 : 0;
}

void ora_closest_hit_batch(const ora_scene *s, int n, const float *o, const float *d, float tmin, float tmax,
                           int32_t *hit, float *t, int32_t *prim, float *b1, float *b2,
                           float *point, float *normal, float *uv) {
#pragma omp parallel for schedule(dynamic, 1024)
    for (int i = 0; i < n; ++i) {
        Hit rec; rec.prim = -1; rec.t = 0; rec.b1 = rec.b2 = 0; rec.point = rec.normal = v3(0); rec.uv = V2{0, 0};
        bool h = closestHit(*s, a3(o + 3 * i), a3(d + 3 * i), tmin, tmax, rec, nullptr);
        hit[i] = h;
        if (!h) { rec.prim = -1; rec.t = 0; rec.b1 = rec.b2 = 0; rec.point = rec.normal = v3(0); rec.uv = V2{0, 0}; }
        if (t) t[i] = rec.t;
        if (prim) prim[i] = rec.prim;
        if (b1) b1[i] = rec.b1;
        if (b2) b2[i] = rec.b2;
        if (point) { point[3 * i] = rec.point.x; point[3 * i + 1] = rec.point.y; point[3 * i + 2] = rec.point.z; }
        if (normal) { normal[3 * i] = rec.normal.x; normal[3 * i + 1] = rec.normal.y; normal[3 * i + 2] = rec.normal.z; }
        if (uv) { uv[2 * i] = rec.uv.x; uv[2 * i + 1] = rec.uv.y; }
    }
}
void ora_any_hit_batch(const ora_scene *s, int n, const float *o, const float *d, const float *tmin,
                       const float *tmax, int32_t *hit) {
#pragma omp parallel for schedule(dynamic, 1024)
    for (int i = 0; i < n; ++i) hit[i] = anyHit(*s, a3(o + 3 * i), a3(d + 3 * i), tmin[i], tmax[i], nullptr);
}

void ora_bxdf_sample_batch(const ora_scene *s, int material, int n, const float *normal, const float *uv,
                           const float *wo, const float *uc, const float *u2,
                           int32_t *ok, float *f, float *wi, float *pdf) {
    const ora_material &m = s->materials[material];
    for (int i = 0; i < n; ++i) {
        BSample bs; bs.f = bs.wi = v3(0); bs.pdf = 0;
        V2 tuv = uv ? V2{uv[2 * i], uv[2 * i + 1]} : V2{0, 0};
        bool r = sampleBxdf(*s, m, a3(normal + 3 * i), tuv, a3(wo + 3 * i), uc[i], V2{u2[2 * i], u2[2 * i + 1]}, bs);
        ok[i] = r;
        if (!r) { bs.f = bs.wi = v3(0); bs.pdf = 0; }
        f[3 * i] = bs.f.x; f[3 * i + 1] = bs.f.y; f[3 * i + 2] = bs.f.z;
        wi[3 * i] = bs.wi.x; wi[3 * i + 1] = bs.wi.y; wi[3 * i + 2] = bs.wi.z;
        pdf[i] = bs.pdf;
    }
}
void ora_bxdf_eval_batch(const ora_scene *s, int material, int n, const float *normal, const float *uv,
                         const float *wo, const float *wi, float *f) {
    const ora_material &m = s->materials[material];
    for (int i = 0; i < n; ++i) {
        V2 tuv = uv ? V2{uv[2 * i], uv[2 * i + 1]} : V2{0, 0};
        V3 r = evalBxdf(*s, m, a3(normal + 3 * i), tuv, a3(wo + 3 * i), a3(wi + 3 * i));
        f[3 * i] = r.x; f[3 * i + 1] = r.y; f[3 * i + 2] = r.z;
    }
}
void ora_bxdf_pdf_batch(const ora_scene *s, int material, int n, const float *normal, const float *uv,
                        const float *wo, const float *wi, float *pdf) {
    const ora_material &m = s->materials[material];
    for (int i = 0; i < n; ++i) {
        V2 tuv = uv ? V2{uv[2 * i], uv[2 * i + 1]} : V2{0, 0};
        pdf[i] = pdfBxdf(*s, m, a3(normal + 3 * i), tuv, a3(wo + 3 * i), a3(wi + 3 * i));
    }
}

void ora_camera_rays(const ora_camera_desc *cam, int n, const int32_t *row, const int32_t *col,
                     const int32_t *sample, float *o, float *d) {
    Cam k = camInit(*cam);
    for (int i = 0; i < n; ++i) {
        Rng rng(row[i], col[i], sample[i] + 1);
        V3 ro, rd; camRay(k, col[i], row[i], sample[i], rng, ro, rd);
        o[3 * i] = ro.x; o[3 * i + 1] = ro.y; o[3 * i + 2] = ro.z;
        d[3 * i] = rd.x; d[3 * i + 1] = rd.y; d[3 * i + 2] = rd.z;
    }
}

void ora_radiance_samples(const ora_scene *s, const ora_camera_desc *cam, int n, const int32_t *row,
                          const int32_t *col, const int32_t *sample, float *rgb) {
    Cam k = camInit(*cam);
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < n; ++i) {
        V3 c = tracePixelSample(*s, k, cam->max_depth, row[i], col[i], sample[i], nullptr);
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
    }
}

// StaticCamera::render camera.cpp:45-128.  32x32 tiles from an atomic job counter; per pixel the samples are
// accumulated in sample order, so the float sums do not depend on the thread count.
void ora_render(const ora_scene *s, const ora_camera_desc *cam, int threads, int sample_begin, int sample_end,
                int reference_barriers, float *acc_rgb, uint8_t *img_rgb, ora_counters *counters) {
    Cam k = camInit(*cam);
    const int W = cam->width, H = cam->height;
    const int spp = cam->x_pixel_samples * cam->y_pixel_samples;
    if (sample_end > spp) sample_end = spp;
    if (threads <= 0) threads = (int) std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    struct Job { int r0, c0, r1, c1; };
    std::vector<Job> jobs;
    for (int r = 0; r < H; r += 32)
        for (int c = 0; c < W; c += 32) jobs.push_back({r, c, std::min(r + 32, H), std::min(c + 32, W)});
    if (sample_begin == 0) std::memset(acc_rgb, 0, sizeof(float) * 3 * (size_t) W * H);
    std::vector<Counters> tc(threads);
    const bool count = counters != nullptr;

    auto tileSamples = [&](const Job &job, int s0, int s1, Counters *cnt) {
        for (int cs = s0; cs < s1; ++cs)
            for (int row = job.r0; row < job.r1; ++row)
                for (int col = job.c0; col < job.c1; ++col) {
                    V3 c = tracePixelSample(*s, k, cam->max_depth, row, col, cs, cnt);
                    float *a = acc_rgb + 3 * ((size_t) row * W + col);
                    a[0] += c.x; a[1] += c.y; a[2] += c.z;                     // image.hpp:82-86
                    if (img_rgb) {
                        float inv = (float) (cs + 1);
                        uint8_t *p = img_rgb + 3 * ((size_t) row * W + col);   // image.hpp:47-52
                        p[0] = toByte(a[0] / inv); p[1] = toByte(a[1] / inv); p[2] = toByte(a[2] / inv);
                    }
                }
    };

    if (reference_barriers) {
        // StaticCamera::render's own schedule (camera.cpp:67-127): threadCount_ PERSISTENT threads pull 32x32 tiles from
        // an atomic job index; one pass = samplesPerPass_ = 1 sample (camera.hpp:181); a std::barrier after every pass
        // whose completion step advances currentSample_ and rewinds the job index (camera.cpp:68-74).
        std::atomic<size_t> next{0};
        std::atomic<int> current{sample_begin};
        std::mutex mu; std::condition_variable cv;
        int waiting = 0; unsigned long generation = 0;
        auto arriveAndWait = [&] {
            std::unique_lock<std::mutex> lk(mu);
            const unsigned long gen = generation;
            if (++waiting == threads) {
                current.fetch_add(1); next.store(0);                            // the barrier's completion function
                waiting = 0; ++generation;
                cv.notify_all();
            } else cv.wait(lk, [&] { return generation != gen; });
        };
        // (a thread that cannot be created -- a container's thread limit -- must not take the process down, and the barrier needs the
        //  exact number of participants: the workers wait at a gate until the pool stands, the barrier then counts the threads that
        //  exist -- ADVICE r3: the earlier probe-then-create left a window in which the limit could change)
        std::atomic<int> gate{0};                                               // 0: wait, 1: go
        std::vector<std::thread> pool;
        struct JoinAll { std::vector<std::thread> &v; std::atomic<int> &g; ~JoinAll() { g.store(1); for (auto &t : v) if (t.joinable()) t.join(); } } ja{pool, gate};
        pool.reserve(threads);
        auto worker = [&](int t) {
            while (gate.load() == 0) std::this_thread::yield();
            while (true) {
                const int cs = current.load();
                if (cs >= sample_end) break;                                    // stopRender_
                while (true) {
                    size_t j = next.fetch_add(1, std::memory_order_relaxed);
                    if (j >= jobs.size()) break;
                    tileSamples(jobs[j], cs, cs + 1, count ? &tc[t] : nullptr);
                }
                arriveAndWait();
            }
        };
        int got = 1;                                                            // the calling thread is participant 0
        for (int t = 1; t < threads; ++t) {
            try { pool.emplace_back(worker, t); ++got; } catch (const std::system_error &) { break; }
        }
        threads = got;                                                          // what arriveAndWait counts to (nobody has passed the gate yet)
        gate.store(1);
        worker(0);
        for (auto &th : pool) th.join();
    } else {
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        struct JoinAll { std::vector<std::thread> &v; ~JoinAll() { for (auto &t : v) if (t.joinable()) t.join(); } } ja{pool};
        pool.reserve(threads);
        auto worker = [&](int t) {
            while (true) {
                size_t j = next.fetch_add(1, std::memory_order_relaxed);
                if (j >= jobs.size()) break;
                tileSamples(jobs[j], sample_begin, sample_end, count ? &tc[t] : nullptr);
            }
        };
        for (int t = 1; t < threads; ++t) {
            try { pool.emplace_back(worker, t); } catch (const std::system_error &) { break; }    // fewer threads: the queue is shared
        }
        worker(0);
        for (auto &th : pool) th.join();
    }
    if (counters) {
        Counters sum;
        for (auto &c : tc) sum.add(c);
        counters->n_camera = sum.n_camera; counters->n_closest = sum.n_closest; counters->n_any = sum.n_any;
        counters->n_nodes_closest = sum.n_nodes_closest; counters->n_tri_closest = sum.n_tri_closest;
        counters->n_accept = sum.n_accept; counters->n_nodes_any = sum.n_nodes_any; counters->n_tri_any = sum.n_tri_any;
        counters->n_shade = sum.n_shade;
        for (int i = 0; i < 8; ++i) { counters->n_shade_class[i] = sum.n_shade_class[i]; counters->n_eval_class[i] = sum.n_eval_class[i]; }
    }
}

} // extern "C"
