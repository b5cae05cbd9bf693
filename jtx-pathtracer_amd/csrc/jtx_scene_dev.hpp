// jtx_scene_dev.hpp -- device scene layout and the BVH traversal / ray-triangle stage.
//
// HBM layout (built once per scene by jtx_mi_scene_create, see DESIGN.md "Data layout"):
//   tnodes: 8 x num_nodes x 2 float4: the reference's 32-B LinearBVHNode (bvh.hpp:7-15) in the near-first
//           depth-first order of each of the 8 direction-sign octants, with a skip link
//           [min.x max.x min.y max.y] [min.z max.z link leaf]  (see traverseThreaded).  A 128-B gfx950
//           cache line holds 4 records, and the record visited next on a hit is the adjacent one.
//   tris  : 3 x float4 per primitive IN BVH ORDER [v0.xyz e1.x] [e1.yz e2.xy] [e2.z - - -]
//           -- Mesh::transform baked, e1 = v1-v0, e2 = v2-v0 precomputed in fp32 exactly as
//              mesh.hpp:109-110 computes them per test, so the per-test index gather + three
//              matrix multiplies of mesh.hpp:71-77 disappear.
//   shade : 4 x float4 per primitive [n0.xyz n1.x] [n1.yz n2.xy] [n2.z uv0.xy uv1.x] [uv1.y uv2.xy asfloat(material)]
//           -- read once per accepted path vertex (the reference interpolates at every accept,
//              mesh.hpp:133-145; only the last one survives, so deferring is equivalent).
// When the 8 orderings + tris fit the LDS budget they are staged into LDS once per workgroup (LdsSrc).
#pragma once
#include "jtx_bxdf.hpp"
#include "jtx_wide_quant.hpp"
#include "jtx_profile.hpp"

namespace jtx {

struct DLight { int type; float position[3]; float intensity[3]; float scale; float scene_radius; int pad[3]; };

struct DevScene {
    const float4    *tnodes;      // threaded (stackless) node records: 8 direction-sign orderings x num_nodes x 2 float4
    const float4    *tris;
    const float4    *shade;
    const DMaterial *materials;
    const DLight    *lights;
    const DTexture  *textures;
    const float     *texels;
    int num_nodes, num_prims, num_lights, num_materials;
    int lds_threaded;      // != 0: the 8 threaded orderings + tris are staged in LDS (stackless kernels)
    int material_mask;     // OR of (1 << Material::type) over the scene's materials
    float sky[3];
    int wide_depth;               // levels of the wide tree = LDS stack entries per lane
    const uint4     *wide;        // 8-ary quantised nodes + leaf records (traverseWide), or null
    // tiny scenes (<= 32 leaves, LDS-resident): the flat leaf list of traverseLeaves
    const float4    *lw_box;      // 2 float4 per leaf, leaves in b.nodes order: [min.x max.x min.y max.y][min.z max.z primitivesOffset numPrimitives]
    const unsigned  *lw_tab;      // per direction-sign octant 16 words: bytes 0..31 = leaf visited at position p, bytes 32..63 = position of leaf l
    int lw_leaves;                // 0: no leaf list
};

struct Counters9 {        // per-lane tallies, reduced per wave (count_rays mode only)
    unsigned n_camera, n_closest, n_any, n_nodes_closest, n_tri_closest, n_accept, n_nodes_any, n_tri_any, n_shade;
    unsigned n_shade_t[8], n_eval_t[8];   // sampleBxdf calls / evalBxdf + pdfBxdf calls by BxDF class (bxdfClass; static indices only: registers)
    JTX_PROF_UTIL_FIELDS         // diagnostic builds only (jtx_profile.hpp); nothing in the product
    JTX_PROF_WIDE_FIELDS
};

// where the per-type tallies live in the scene's 64-word counter block: [0..8] the nine ray counters, [9..53] diagnostics, then these
constexpr int CNT_SHADE_T = 32, CNT_EVAL_T = 52;
// BxDF class of a material = the code path its sampleBxdf / evalBxdf / pdfBxdf calls take (jtx_mi.h: jtx_mi_counters):
// 0 DIFFUSE, 1 DIELECTRIC rough, 2 CONDUCTOR rough, 3 METALLIC_ROUGHNESS, 4 ThinDielectric, 5 DIELECTRIC smooth or index-matched
// (dielectric.hpp:44: eta == 1 || smooth), 6 CONDUCTOR smooth (conductor.hpp:33; smooth = max(alpha) < 1e-3, microfacet.hpp:23-25)
JD int bxdfClass(const DMaterial &m) {
    const bool smooth = fmax2(m.alpha_x, m.alpha_y) < 1e-3f;
    if (m.type == 1) return (m.ior[0] == 1.0f || smooth) ? 5 : 1;
    if (m.type == 2) return smooth ? 6 : 2;
    return m.type;
}
JD void countClass(unsigned (&a)[8], int c) { a[0] += c == 0; a[1] += c == 1; a[2] += c == 2; a[3] += c == 3; a[4] += c == 4; a[5] += c == 5; a[6] += c == 6; }

enum { SRC_GLOBAL = 0, SRC_LDS = 1, SRC_WIDE = 2, SRC_LEAF = 3 };

struct GlobalSrc {
    const float4 *tnodes, *tris;
    JD float4 tnode(int i, int h) const { return tnodes[2 * i + h]; }
    JD float4 tri(int i, int h) const { return tris[3 * i + h]; }
};
// The workgroup's LDS copy.  The two 16-B halves of the records are kept as two arrays ([all first halves][all second
// halves]): a ds_read_b128 serves 16 lanes at a time over 64 banks, and with interleaved 32-B records every first half
// starts on an even bank quad -- 8 positions for 16 lanes, a 2-way conflict at best; split, a half of record i sits on
// quad i mod 16 (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the C2 kernel: 28 % interleaved).
// The same for the triangles (JTX_LDS_PLANES): three float4 planes [all first granules][all second][all third] -- record i of a plane on
// bank quad i mod 16 -- instead of 48-byte records (granule h of triangle i on quad (3 i + h) mod 16).
#ifndef JTX_LDS_PLANES
#define JTX_LDS_PLANES 1
#endif
struct LdsSrc {
    const float4 *tnodes, *tris;
    int half;             // records per half array = 8 * num_nodes
    int np;               // triangles (plane stride)
    JD float4 tnode(int i, int h) const { return tnodes[h * half + i]; }
#if JTX_LDS_PLANES
    JD float4 tri(int i, int h) const { return tris[h * np + i]; }
#else
    JD float4 tri(int i, int h) const { return tris[3 * i + h]; }
#endif
};

struct HitRec { float t; int prim; float b1, b2; };

// Moller-Trumbore, Mesh::tClosestHit / tAnyHit (mesh.hpp:106-127, 168-192) on the pre-baked record.
//  |det| < 1e-8 is a double compare in the reference (mesh.hpp:114); for a float |det| it is
//  equivalent to |det| <= 1e-8f because 1e-8f < 1e-8 < nextafter(1e-8f).
template <class Src>
JD bool triTest(const Src &src, int prim, f3 o, f3 d, float tmin, float tmax, float &b1, float &b2, float &root) {
    const float4 q0 = src.tri(prim, 0), q1 = src.tri(prim, 1), q2 = src.tri(prim, 2);
    const f3 v0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q0.w, q1.x, q1.y), e2 = mk3(q1.z, q1.w, q2.x);
    const f3 pvec = cross(d, e2);
    const float det = dot(e1, pvec);
    if (fabsf(det) <= 1e-8f) return false;
    const float invDet = 1.0f / det;
    const f3 tvec = o - v0;
    b1 = dot(tvec, pvec) * invDet;
    if (b1 < 0.0f || b1 > 1.0f) return false;
    const f3 qvec = cross(tvec, e1);
    b2 = dot(d, qvec) * invDet;
    if (b2 < 0.0f || b1 + b2 > 1.0f) return false;
    root = dot(e2, qvec) * invDet;
    return tmin < root && root < tmax;
}

// AABB::hit (aabb.hpp:66-81) exactly as written, for rays the fast form below cannot take.
//  * 1/d is computed once per ray: aabb.hpp:71 recomputes the same quotient at every node.
//  * The per-axis early-out of aabb.hpp:78 is folded into one test after the third axis: t0 only
//    grows and t1 only shrinks (a NaN candidate is never selected by either ternary), so
//    "t0 > t1 after some axis" <=> "t0 > t1 after the last axis".
JD bool slabExact(const float4 na, const float4 nb, f3 o, f3 inv, float t0, float t1) {
    {
        float tn = (na.x - o.x) * inv.x, tf = (na.y - o.x) * inv.x;
        float lo = tn > tf ? tf : tn, hi = tn > tf ? tn : tf;
        t0 = lo > t0 ? lo : t0; t1 = hi < t1 ? hi : t1;
    }
    {
        float tn = (na.z - o.y) * inv.y, tf = (na.w - o.y) * inv.y;
        float lo = tn > tf ? tf : tn, hi = tn > tf ? tn : tf;
        t0 = lo > t0 ? lo : t0; t1 = hi < t1 ? hi : t1;
    }
    {
        float tn = (nb.x - o.z) * inv.z, tf = (nb.y - o.z) * inv.z;
        float lo = tn > tf ? tf : tn, hi = tn > tf ? tn : tf;
        t0 = lo > t0 ? lo : t0; t1 = hi < t1 ? hi : t1;
    }
    return !(t0 > t1);
}

// The same test for a REGULAR ray: every 1/d finite and non-zero, origin finite, interval not NaN.
// Then tNear/tFar are never NaN (no 0*inf, no inf-inf), so the reference's compare-and-swap is
// min/max of the pair and its running t0/t1 ternaries are max/min chains (associative up to the sign
// of zero, which only ever feeds the comparison t0 > t1).  17 VALU instead of ~40.
JD bool slabRegular(const float4 na, const float4 nb, f3 o, f3 inv, float tmin, float tmax) {
    const float ax = (na.x - o.x) * inv.x, bx = (na.y - o.x) * inv.x;
    const float ay = (na.z - o.y) * inv.y, by = (na.w - o.y) * inv.y;
    const float az = (nb.x - o.z) * inv.z, bz = (nb.y - o.z) * inv.z;
    const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
    const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    return t0 <= t1;
}

// Phase A of the flat leaf list, the near / far planes of an axis picked by the SIGN of 1/d instead of by min / max (v_min_f32 / v_max_f32
// issue at 4.4 cycles, v_mul_f32 / v_fma_f32 at 2.4: profiles/r03_valu_rates.txt).  ip = 1/d where it is positive and 0 elsewhere, in = 1/d
// where it is negative and 0 elsewhere; a = lo - o <= b = hi - o.  For 1/d > 0: fma(a, ip, b * in) = a / d + (+-0) = a / d = min(a / d, b / d)
// (monotone rounding), fma(b, ip, a * in) = b / d = the max; for 1/d < 0 the roles swap.  Equal to slabRegular's operands bit for bit up to
// the sign of a zero, which only ever reaches the comparison t0 <= t1.
template <bool OPEN>
JD bool slabRegularSel(const float4 na, const float4 nb, f3 o, f3 ip, f3 in, float tmin, float tmax) {
    const float ax = na.x - o.x, bx = na.y - o.x;
    const float ay = na.z - o.y, by = na.w - o.y;
    const float az = nb.x - o.z, bz = nb.y - o.z;
    const float nx = __builtin_fmaf(ax, ip.x, bx * in.x), fx = __builtin_fmaf(bx, ip.x, ax * in.x);
    const float ny = __builtin_fmaf(ay, ip.y, by * in.y), fy = __builtin_fmaf(by, ip.y, ay * in.y);
    const float nz = __builtin_fmaf(az, ip.z, bz * in.z), fz = __builtin_fmaf(bz, ip.z, az * in.z);
    const float t0 = fmaxf(fmaxf(nx, ny), fmaxf(nz, tmin));
    const float t1 = OPEN ? fminf(fminf(fx, fy), fz) : fminf(fminf(fx, fy), fminf(fz, tmax));
    return t0 <= t1;
}

JD bool finiteNonZero(float x) { return fabsf(x) < __builtin_inff() && x != 0.0f; }

// Scene::closestHit (scene.cpp:10-55) / Scene::anyHit (scene.cpp:57-94).
// "while-while" form: every lane first walks interior nodes until it stands on a leaf whose box it
// hits (or its stack runs dry), then the wave tests leaf triangles together -- the visit ORDER per
// ray, the near-child-first rule (dirIsNeg[axis], scene.cpp:40-46) and the shrinking t.max are
// exactly the reference's; only the interleaving between lanes differs.
//  * stack: one LDS column per lane (stk[level * stride]), depth bounded by the BVH build.
#ifndef JTX_STEPS_PER_VOTE
#define JTX_STEPS_PER_VOTE 4     // interior steps between two scheduling votes (the ballots are pure overhead)
#endif
#ifndef JTX_LEAF_VOTE
#define JTX_LEAF_VOTE 12      // lanes parked on a leaf that end the interior phase of a wave
#endif
#ifndef JTX_FEW_WALKERS
#define JTX_FEW_WALKERS 12    // ... or any parked lane once so few lanes are still walking (C2: 47.3 -> 45.6 ms)
#endif

// ---- threaded (stackless) traversal -------------------------------------------------------------------
// In Scene::closestHit / anyHit the child visited first depends only on the SIGN of the ray direction
// along the node's split axis (dirIsNeg[axis], scene.cpp:11-12,40-46).  For a fixed sign octant the
// near-first depth-first order of the whole tree is therefore a fixed sequence, and "pop the stack"
// always means "jump behind the current subtree".  scene_create lays the nodes out once per octant in
// that order (8 x num_nodes records of 32 B) with a skip link:
//     interior: [box][skip = index of the first record behind this subtree, or -1][0]
//     leaf    : [box][primitivesOffset][numPrimitives | 0x80000000 if it is the last record]
// and the traversal becomes: test the box; hit or leaf -> next record, miss on an interior -> skip.
// Same node visits, same order, same box tests against the same shrinking t.max as the reference's
// stack machine -- but no stack, no push/pop branches, and ~40 % fewer instructions per node.
template <bool ANY, bool COUNT, bool REGULAR, class Src>
JD bool traverseThreaded(const Src &src, int num_nodes, f3 o, f3 d, f3 inv, int negmask, float tmin, float tmax,
                         HitRec &rec, Counters9 &cnt) {
    int cur = negmask * num_nodes;                       // root record of this ray's octant
    int leafW = 0, leafOff = 0;                          // leafW != 0: parked on a leaf (count | last flag)
    bool hitAnything = false;
    if (COUNT) { if (ANY) cnt.n_any++; else cnt.n_closest++; }
    UTIL(if (COUNT) cnt.it_calls++;)
    while (true) {
        while (true) {
#pragma unroll
            for (int rep = 0; rep < JTX_STEPS_PER_VOTE; ++rep) {
                UTIL(if (COUNT) { cnt.it_interior++; const int na = __popcll(__ballot(leafW == 0 && cur >= 0));
                                  cnt.it_hist[na <= 2 ? 0 : na <= 4 ? 1 : na <= 8 ? 2 : na <= 16 ? 3 : na <= 32 ? 4 : na <= 48 ? 5 : 6]++;
                                  if (leafW != 0) cnt.it_np++; else if (cur < 0) cnt.it_nd++; })
                if (leafW == 0 && cur >= 0) {
                    const float4 na = src.tnode(cur, 0);
                    const float4 nb = src.tnode(cur, 1);
                    if (COUNT) { if (ANY) cnt.n_nodes_any++; else cnt.n_nodes_closest++; }
                    const bool boxHit = REGULAR ? slabRegular(na, nb, o, inv, tmin, tmax) : slabExact(na, nb, o, inv, tmin, tmax);
                    const int w = __float_as_int(nb.w), z = __float_as_int(nb.z);
                    if (boxHit && w != 0) { leafW = w; leafOff = z; }                 // park on the leaf
                    else cur = (boxHit || w != 0) ? (w < 0 ? -1 : cur + 1) : z;       // next record / skip link
                }
            }
            const unsigned long long walking = __ballot(leafW == 0 && cur >= 0);
            const unsigned long long parked = __ballot(leafW != 0);
            if (walking == 0ull || __popcll(parked) >= JTX_LEAF_VOTE) break;
            if (parked != 0ull && __popcll(walking) <= JTX_FEW_WALKERS) break;      // do not let a few long walks hold the parked lanes
        }
        if (__ballot(leafW != 0) == 0ull) break;            // wave-uniform: every lane is done
        UTIL(if (COUNT) { cnt.it_leaf++; if (leafW == 0) { if (cur < 0) cnt.it_ld++; else cnt.it_lw++; } })
        if (leafW != 0) {
            const int n = leafW & 0xffff;
            for (int i = 0; i < n; ++i) {
                const int prim = leafOff + i;
                if (COUNT) { if (ANY) cnt.n_tri_any++; else cnt.n_tri_closest++; }
                float b1, b2, root;
                if (!triTest(src, prim, o, d, tmin, tmax, b1, b2, root)) continue;
                hitAnything = true;
                if (ANY) break;
                tmax = root;
                rec.t = root; rec.prim = prim; rec.b1 = b1; rec.b2 = b2;
                if (COUNT) cnt.n_accept++;
            }
            cur = ((ANY && hitAnything) || leafW < 0) ? -1 : cur + 1;
            leafW = 0;
        }
    }
    return hitAnything;
}

// ---- wide (8-ary, quantised) traversal for HBM-resident scenes ---------------------------------------
// What the reference computes for a REGULAR ray (see slabRegular) does not depend on the interior
// nodes at all: boxes nest exactly (a node's box is the union of its children's, initBranch bvh.hpp:31-37) and
// every operation of the slab test is monotone in the box planes and in t.max, so "the leaf's own box
// test passes" implies "every ancestor's test passed" -- earlier, with a t.max that was no smaller.
// closestHit / anyHit are therefore:  walk the LEAVES in the octant's fixed near-first order; test the
// leaf's box against the current interval; if it passes, test its triangles.  Any structure that walks
// the leaves in that order and only skips leaves whose box test would fail gives bit-identical hits.
// The uncounted kernels use that freedom (the counted ones keep the reference's node visits):
//   wide node (192 B = 12 x 16 B): a treelet of the binary tree below a node cut into at most 8
//     children (the cut that minimises the summed box area of all wide nodes, jtx_capi.hip); slots = its wide (interior)
//     children first, then its leaves, each left to right; every child box quantised OUTWARD to 8 bits per plane on the
//     node's own grid (plane = origin + q 2^e, checked in exact arithmetic by the builder), so a slab test on it can only
//     pass more often than on any exact box inside it.  Bit layout, children blocks and the root-peel record:
//     jtx_wide_quant.hpp (the one place that encodes them).
//       [origin.xyz | ex ey ez, #interior, #children] [lo.x x8 | lo.y x8] [lo.z x8 | hi.x x8] [hi.y x8 | hi.z x8]
//       8 x [children base | the octant's 24-bit visiting order | one-hot position of slots 0-3 | of slots 4-7]
//     visiting order of an octant: the slots in the order the reference's near-first rule (dirIsNeg[axis], scene.cpp:40-46)
//     walks the treelet, 3 bits per position.  A ray reads the tail of ITS octant; the one-hot words turn the eight pass / miss
//     bytes of the box tests (sign bytes of t1 - t0, v_perm_b32) into the pending mask of the order list with one AND and one
//     byte sum (wideNodePend) -- rounds 1-3 permuted the eight hit bits with 26 instructions.
//   leaf record (32 B): the exact leaf box + primitivesOffset + numPrimitives, tested with slabRegular.
//   (The head of the array holds the exact boxes of the root's children -- the "root-peel" record of a first step through scalar
//    loads that was measured and dropped; tests walk it.  That step, the one- and two-tail node formats, the straggler-slip kernel and
//    the lane-column stack addressing live as patches under tools/experiments/, not here.)
// The per-lane stack holds one 64-bit entry per wide level {header word, visiting order, pending positions} in LDS.
// A stale hit bit (t.max shrank since the node was tested) only costs a visit.  Irregular rays (a zero / non-finite
// direction component ...) take the exact binary path.
constexpr unsigned WIDE_NODE_G = jtxq::kNodeG;   // granules (16 B) from one interior child to the next
#ifndef JTX_WIDE_LEAF_VOTE
#define JTX_WIDE_LEAF_VOTE 16
#endif
#ifndef JTX_WIDE_FEW_WALKERS
#define JTX_WIDE_FEW_WALKERS 8          // (re-swept on the per-octant tails: 4 / 8 / 12 / 16 / 24 -> C3 300.8 / 299.8 / 303.0 / 304.5 / 309.0 ms, C5 +-1)
#endif
#ifndef JTX_WIDE_STEPS
#define JTX_WIDE_STEPS 1
#endif
#ifndef JTX_WIDE_PEND_BARRIER
#define JTX_WIDE_PEND_BARRIER 1
#endif
#ifndef JTX_FP_TOLERANCE
#define JTX_FP_TOLERANCE 0           // 1: the measurement build of round 5 (what north_star's "stated per-pixel tolerance" would buy in the traversal)
#endif

// outward slack of the wide-node slab test: mu = 2^-23 (4 |b| + 512 |a|) + 2^-100 per axis (error budget in DESIGN.md)
#define WIDE_MU_B 4.76837158203125e-07f
#define WIDE_MU_A 6.103515625e-05f
#define WIDE_MU_0 7.888609052210118e-31f
constexpr float WIDE_RANGE = 1099511627776.0f;   // 2^40: |1/d|, 1/|1/d| and |o| of a ray the wide nodes may take (no overflow, a exact)

JD float ubyteToFloat(unsigned v, int k) { return (float) ((v >> (8 * k)) & 0xffu); }   // v_cvt_f32_ubyteK

struct WideRay { f3 o, d, inv; float tmin, tmax; int negmask; };
struct WideState {
    unsigned gbase, gbits;       // current group: children base (28 bits) | interior children (4) ; order list (24) | pending (8)
    int sp, pendLeaf;            // stack entries in use; granule of the leaf record the lane is parked on (-1: none)
    bool done, hitAnything;
    // group = the children of one wide node still to visit; the start group is "the root": a block of one interior child at kRootNode
    JD void start() { gbase = jtxq::groupWord(jtxq::kRootNode, 1); gbits = 1u; sp = 0; pendLeaf = -1; done = false; hitAnything = false; }
    JD bool walking() const { return pendLeaf < 0 && !done; }
};

// The 8 child boxes of a wide node against a ray (never misses a box AABB::hit would pass), with the answer in POSITION space of the
// ray's order list: ohLo / ohHi hold, per slot byte, the one-hot
// position of the slot in the ray's visiting order (0: no child).  Per child a DIFFERENCE instead of a compare (a miss is a NEGATIVE
// difference: near and far are finite for the rays the wide nodes take, far is never -0 -- every far value is b + mu with mu > 0 -- and
// equal operands give +0; since round 5 the smallest of far - near, t.max - near and far - t.min, see below), v_perm_b32's sign selectors turn four sign bits into four bytes 0xff / 0x00, (not miss) AND one-hot, summed
// over the bytes (distinct bits: a sum is an OR), is the pending mask: 8 v_sub + 4 v_perm + 3 v_bitop3 + 1 v_sad_u8 (48 cycles by
// tools/micro/rate7.hip) where compare / select / or and the 8-bit permutation took 45 instructions of the 4.4-cycle class (198).
JD unsigned wideNodePend(const uint4 n0, const uint4 n2, const uint4 n3, const uint4 n4, unsigned ohLo, unsigned ohHi, f3 o, f3 inv, float tmin, float tmax) {
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    const float axx = __uint_as_float((n0.w & 0xffu) << 23) * inv.x, bxx = (__uint_as_float(n0.x) - o.x) * inv.x;
    const float ayy = __uint_as_float(((n0.w >> 8) & 0xffu) << 23) * inv.y, byy = (__uint_as_float(n0.y) - o.y) * inv.y;
    const float azz = __uint_as_float(((n0.w >> 16) & 0xffu) << 23) * inv.z, bzz = (__uint_as_float(n0.z) - o.z) * inv.z;
    const float mux = __fmaf_rn(fabsf(bxx), WIDE_MU_B, __fmaf_rn(fabsf(axx), WIDE_MU_A, WIDE_MU_0));
    const float muy = __fmaf_rn(fabsf(byy), WIDE_MU_B, __fmaf_rn(fabsf(ayy), WIDE_MU_A, WIDE_MU_0));
    const float muz = __fmaf_rn(fabsf(bzz), WIDE_MU_B, __fmaf_rn(fabsf(azz), WIDE_MU_A, WIDE_MU_0));
    const float bnx = bxx - mux, bfx = bxx + mux, bny = byy - muy, bfy = byy + muy, bnz = bzz - muz, bfz = bzz + muz;
    const unsigned nxq[2] = {nx ? n3.z : n2.x, nx ? n3.w : n2.y}, fxq[2] = {nx ? n2.x : n3.z, nx ? n2.y : n3.w};
    const unsigned nyq[2] = {ny ? n4.x : n2.z, ny ? n4.y : n2.w}, fyq[2] = {ny ? n2.z : n4.x, ny ? n2.w : n4.y};
    const unsigned nzq[2] = {nz ? n4.z : n3.x, nz ? n4.w : n3.y}, fzq[2] = {nz ? n3.x : n4.z, nz ? n3.y : n4.w};
    float df[8];
    unsigned mk[4];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int w = s >> 2, b = s & 3;
        // the interval's ends as DIFFERENCES (v_sub_f32 issues beside the v_min / v_max class, profiles/r05_box_rates.txt): the child is missed
        // iff far < near or t.max < near or far < t.min -- min(far, t.max) < max(near, t.min) without its "t.max < t.min" term (an empty
        // interval then only costs visits)
        const float tn = fmaxf(fmaxf(__fmaf_rn(ubyteToFloat(nxq[w], b), axx, bnx), __fmaf_rn(ubyteToFloat(nyq[w], b), ayy, bny)), __fmaf_rn(ubyteToFloat(nzq[w], b), azz, bnz));
        const float tf = fminf(fminf(__fmaf_rn(ubyteToFloat(fxq[w], b), axx, bfx), __fmaf_rn(ubyteToFloat(fyq[w], b), ayy, bfy)), __fmaf_rn(ubyteToFloat(fzq[w], b), azz, bfz));
        df[s] = fminf(fminf(tf - tn, tmax - tn), tf - tmin);
#if defined(__HIP_DEVICE_COMPILE__)
        // v_perm_b32 D, S0, S1, sel: selector 0x0b = 8 x S0[31], 0x09 = 8 x S1[31], 0x0c = 0x00 (checked on the chip: rate7.hip)
        if (s & 1) {
            mk[s >> 1] = __builtin_amdgcn_perm(__float_as_uint(df[s - 1]), __float_as_uint(df[s]), (s & 2) ? 0x090b0c0cu : 0x0c0c090bu);
#if JTX_WIDE_PEND_BARRIER
            __builtin_amdgcn_sched_barrier(0);                  // two children at a time: eight differences live at once spill INSIDE the node loop
#endif
        }
#endif
    }
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned xl = __builtin_amdgcn_bitop3_b32(mk[0], mk[1], ohLo, 0x02);          // ~a & ~b & c
    const unsigned xh = __builtin_amdgcn_bitop3_b32(mk[2], mk[3], ohHi, 0x02);
    return __builtin_amdgcn_sad_u8(__builtin_amdgcn_bitop3_b32(xl, xh, xh, 0xfc), 0u, 0u);   // a | b
#else
    unsigned pend = 0u;
    for (int s = 0; s < 8; ++s) if (!(df[s] < 0.0f) && !(__float_as_uint(df[s]) >> 31)) pend |= ((s < 4 ? ohLo : ohHi) >> (8 * (s & 3))) & 0xffu;
    return pend;
#endif
}

// One interior step of a walking lane: take the next child of the current group (popping the stack when the
// group is empty); an interior child is fetched and tested (its hits become the new group), a leaf child parks the lane.
// ORD = 0 (anyHit: the answer does not depend on the order): children are taken in slot order, leaves first; 1: the octant's visiting order.
template <int ORD>
JD void wideNodeStep(const uint4 *__restrict__ wide, uint2 *stk, int stride, const WideRay &r, WideState &ws) {
    constexpr bool ORDERED = ORD == 1;
    if ((ws.gbits & 0xffu) == 0u) {                          // group exhausted: pop
        if (ws.sp == 0) { ws.done = true; return; }
        --ws.sp; const uint2 e = stk[ws.sp * stride]; ws.gbase = e.x; ws.gbits = e.y;
    }
    // next position in visiting order (anyHit: from the end of the slot list -- leaves first)
    const int k = ORDERED ? __builtin_ctz(ws.gbits & 0xffu) : 31 - __builtin_clz(ws.gbits & 0xffu);
    ws.gbits &= ~(1u << k);
    const unsigned slot = (ws.gbits >> (8 + 3 * k)) & 7u;
    const unsigned ni = ws.gbase >> 28, base = ws.gbase & 0x0fffffffu;
    if (slot >= ni) { ws.pendLeaf = (int) (base + WIDE_NODE_G * ni + 2u * (slot - ni)); return; }
    const unsigned a = base + WIDE_NODE_G * slot;
    if (ws.gbits & 0xffu) { stk[ws.sp * stride] = make_uint2(ws.gbase, ws.gbits); ++ws.sp; }
    const uint4 n0 = wide[a], n2 = wide[a + 1], n3 = wide[a + 2], n4 = wide[a + 3];
    // tail of the ray's octant: [children base | its order | one-hot positions of slots 0-3 | 4-7]
    const uint4 tl = wide[a + 4 + (ORDERED ? (unsigned) r.negmask : 0u)];
    const unsigned nchild = n0.w >> 28;
    const unsigned perm = ORDERED ? tl.y : 0x00fac688u;            // (anyHit: the identity list, slot k at position k)
    const unsigned pend = ORDERED ? wideNodePend(n0, n2, n3, n4, tl.z, tl.w, r.o, r.inv, r.tmin, r.tmax)
                                  : wideNodePend(n0, n2, n3, n4, 0x08040201u, 0x80402010u, r.o, r.inv, r.tmin, r.tmax) & ((1u << nchild) - 1u);
    ws.gbase = tl.x | (((n0.w >> 24) & 0xfu) << 28);
    ws.gbits = pend | (perm << 8);
}

// The leaf a lane is parked on: AABB::hit on the exact box, then the leaf's triangles (mesh.hpp:106-192)
template <class Src>
JD void wideLeafStep(const uint4 *__restrict__ wide, const Src &src, bool any, WideRay &r, WideState &ws, HitRec &rec) {
    const uint4 ua = wide[ws.pendLeaf], ub = wide[ws.pendLeaf + 1];
    const float4 la = make_float4(__uint_as_float(ua.x), __uint_as_float(ua.y), __uint_as_float(ua.z), __uint_as_float(ua.w));
    const float4 lb = make_float4(__uint_as_float(ub.x), __uint_as_float(ub.y), 0.0f, 0.0f);
#if JTX_FP_TOLERANCE
    // tolerance build (DESIGN.md "tolerance mode"; NOT the product): the leaf's exact box is not re-tested -- its triangles are, so a hit
    // can only differ from the reference's where AABB::hit's rounding rejects a box whose triangle test would pass (grazing hits)
    (void) la; (void) lb;
    {
#else
    if (slabRegular(la, lb, r.o, r.inv, r.tmin, r.tmax)) {
#endif
        const int n = (int) ub.w, off = (int) ub.z;
        for (int i = 0; i < n; ++i) {
            const int prim = off + i;
            float b1, b2, root;
            if (!triTest(src, prim, r.o, r.d, r.tmin, r.tmax, b1, b2, root)) continue;
            ws.hitAnything = true;
            if (any) break;
            r.tmax = root;
            rec.t = root; rec.prim = prim; rec.b1 = b1; rec.b2 = b2;
        }
        if (any && ws.hitAnything) ws.done = true;
    }
    ws.pendLeaf = -1;
}

JD void wideRaySetup(WideRay &r, f3 o, f3 d, f3 inv, int negmask, float tmin, float tmax) {
    r.o = o; r.d = d; r.inv = inv; r.tmin = tmin; r.tmax = tmax; r.negmask = negmask;
}

template <bool ANY, int STEPS, class Src>
JD bool traverseWide(const uint4 *__restrict__ wide, const Src &src, uint2 *stk, int stride, f3 o, f3 d, f3 inv,
                     int negmask, float tmin, float tmax, HitRec &rec, Counters9 &cnt) {
    WideRay r; wideRaySetup(r, o, d, inv, negmask, tmin, tmax);
    WideState ws;
    ws.start();
    WSTAT(cnt.w_calls++;)
    while (true) {
        while (true) {
#pragma unroll
            for (int rep = 0; rep < STEPS; ++rep) {
                WSTAT(cnt.w_node_iters++;
                      { const int na = __popcll(__ballot(ws.walking()));
                        cnt.w_hist[na <= 2 ? 0 : na <= 4 ? 1 : na <= 8 ? 2 : na <= 16 ? 3 : na <= 32 ? 4 : na <= 48 ? 5 : 6]++; })
                WSTAT(if (ws.pendLeaf >= 0) cnt.w_np++; else if (ws.done) cnt.w_nd++;)
                if (ws.walking()) {
                    WSTAT(cnt.w_node_steps++; const int leafBefore = ws.pendLeaf; const bool doneBefore = ws.done;)
                    wideNodeStep<ANY ? 0 : 1>(wide, stk, stride, r, ws);
                    WSTAT(if (ws.pendLeaf < 0 && !ws.done) cnt.w_fetch++; (void) leafBefore; (void) doneBefore;)
                }
            }
            const unsigned long long walking = __ballot(ws.walking());
            const unsigned long long parked = __ballot(ws.pendLeaf >= 0);
            if (walking == 0ull || __popcll(parked) >= JTX_WIDE_LEAF_VOTE) break;
            if (parked != 0ull && __popcll(walking) <= JTX_WIDE_FEW_WALKERS) break;
        }
        if (__ballot(ws.pendLeaf >= 0) == 0ull) break;          // wave-uniform: every lane is done
        WSTAT(cnt.w_leaf_iters++; if (ws.pendLeaf < 0) { if (ws.done) cnt.w_ld++; else cnt.w_lw++; })
        if (ws.pendLeaf >= 0) { WSTAT(cnt.w_leaf_steps++;) wideLeafStep(wide, src, ANY, r, ws, rec); }
    }
    return ws.hitAnything;
}

JD bool wideRayOk(f3 o, f3 inv, float tmin, float tmax);

JD bool regularRay(f3 o, f3 inv, float tmin, float tmax) {
    return finiteNonZero(inv.x) && finiteNonZero(inv.y) && finiteNonZero(inv.z) &&
           fabsf(o.x) < __builtin_inff() && fabsf(o.y) < __builtin_inff() && fabsf(o.z) < __builtin_inff() &&
           tmin == tmin && tmax == tmax;
}

// rays the wide nodes may take: regular, and in the range where cell / d is exact and nothing overflows
JD bool wideRayOk(f3 o, f3 inv, float tmin, float tmax) {
    const float hi = fmaxf(fmaxf(fabsf(inv.x), fabsf(inv.y)), fabsf(inv.z)), lo = fminf(fminf(fabsf(inv.x), fabsf(inv.y)), fabsf(inv.z));
    return regularRay(o, inv, tmin, tmax) && hi <= WIDE_RANGE && lo >= 1.0f / WIDE_RANGE &&
           fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) <= WIDE_RANGE;
}

// HBM-resident scene, uncounted kernels: wide traversal; a wave with an irregular ray walks the binary records
// SC / SA: node steps between two scheduling votes of a closestHit / anyHit walk (the votes are pure overhead, a second step before the
// vote lets a finished lane idle one step longer: the Lambert kernel gains with 2, the all-BxDF kernel -- more registers live -- loses)
template <int SC, int SA>
struct WideSrcT {
    static constexpr int stepsClosest = SC, stepsAny = SA;
    const uint4 *wide;
    const float4 *tnodes, *tris;
    uint2 *stk;               // this lane's LDS stack column
    int stride;               // entries between two levels (= workgroup size)
    JD float4 tnode(int i, int h) const { return tnodes[2 * i + h]; }
    JD float4 tri(int i, int h) const { return tris[3 * i + h]; }
};
typedef WideSrcT<JTX_WIDE_STEPS, JTX_WIDE_STEPS> WideSrc;

template <bool ANY, bool COUNT, int SC, int SA>
JD bool traverseNoStack(const WideSrcT<SC, SA> &src, int num_nodes, f3 o, f3 d, float tmin, float tmax, HitRec &rec, Counters9 &cnt) {
    static_assert(!COUNT, "the counted kernels reproduce the reference's node visits: binary records only");
    if (num_nodes == 0) return false;
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const int negmask = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
    if (__builtin_expect(__ballot(!wideRayOk(o, inv, tmin, tmax)) == 0ull, 1))
        return traverseWide<ANY, (ANY ? SA : SC)>(src.wide, src, src.stk, src.stride, o, d, inv, negmask, tmin, tmax, rec, cnt);
    return traverseThreaded<ANY, false, false>(src, num_nodes, o, d, inv, negmask, tmin, tmax, rec, cnt);
}

// stackless entry point (k_render_pixels and the per-ray test kernels)
template <bool ANY, bool COUNT, class Src>
JD bool traverseNoStack(const Src &src, int num_nodes, f3 o, f3 d, float tmin, float tmax, HitRec &rec, Counters9 &cnt) {
    if (num_nodes == 0) return false;
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const int negmask = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
    // (== regularRay(); spelled out because calling the helper here costs the LDS-resident kernel 11 more spilled
    //  VGPRs and 3 % of its time with this compiler -- measured)
    const bool regular = finiteNonZero(inv.x) && finiteNonZero(inv.y) && finiteNonZero(inv.z) &&
                         fabsf(o.x) < __builtin_inff() && fabsf(o.y) < __builtin_inff() && fabsf(o.z) < __builtin_inff() &&
                         tmin == tmin && tmax == tmax;
    if (__builtin_expect(regular, 1)) return traverseThreaded<ANY, COUNT, true>(src, num_nodes, o, d, inv, negmask, tmin, tmax, rec, cnt);
    return traverseThreaded<ANY, COUNT, false>(src, num_nodes, o, d, inv, negmask, tmin, tmax, rec, cnt);   // axis-parallel & co.
}

// ---- flat leaf list for tiny scenes (uncounted kernels, <= 32 leaves, everything in LDS) ---------------------------
// For a REGULAR ray the reference's result is a function of the LEAVES alone (see the wide traversal above): walk the
// leaves in the octant's near-first order; test the leaf's own box against the current interval; if it passes, test its
// triangles.  With a few dozen leaves the interior nodes are not worth their divergence -- a wave sits through 38
// interior iterations for 15.7 node visits per ray on the Cornell box, a third of the lanes working -- so:
//   phase A  every leaf's box against the ray's INITIAL interval, leaf by leaf, the same leaf for all 64 lanes: no
//            divergence, no votes, boxes as scalar operands; the passes land in a 32-bit candidate mask, bit = the leaf's
//            position in this octant's visiting order.  (t.max only shrinks and the slab test is monotone in it: the
//            leaves that pass later form a subset.)
//   phase B  the candidates in visiting order; once the lane has accepted a hit (t.max shrank) a candidate's box is
//            re-tested against the current interval -- the reference's own leaf test -- before its triangles.
// Same leaves tested against the same intervals in the same order as Scene::closestHit / anyHit => same hits bit for bit.
#ifndef JTX_LEAF_TRI_VOTE
#define JTX_LEAF_TRI_VOTE 1     // lanes parked on a leaf that end the candidate walk (C2: 1 -> 27.9 ms, 8 -> 28.4, 24 -> 29.1, 40 -> 29.4)
#endif
struct LeafSrc {
    const float4 *tnodes, *tris; int half;          // the LDS copy of the binary records (irregular rays) and triangles
    const float4 *lbox;                             // LDS copy of the leaf list
    const float4 *gbox;                             // the same in HBM: read with wave-uniform indices -> scalar loads
    const unsigned *tab;                            // LDS copy of the order / position tables
    int nleaf;
    int np, lpad;                                   // triangles (plane stride); padded leaf count (stride between the halves of lbox)
    const float4 *groot = nullptr;                  // the binary records in HBM (record 0 = the root's box), or null
    bool fresh = false;                             // this lane's ray is a camera ray (path kernel: depth 0)
    JD float4 tnode(int i, int h) const { return tnodes[h * half + i]; }
#if JTX_LDS_PLANES
    JD float4 tri(int i, int h) const { return tris[h * np + i]; }
    JD float4 leafbox(int leaf, int h) const { return lbox[h * lpad + leaf]; }
#else
    JD float4 tri(int i, int h) const { return tris[3 * i + h]; }
    JD float4 leafbox(int leaf, int h) const { return lbox[2 * leaf + h]; }
#endif
};

template <bool ANY>
JD bool traverseLeaves(const LeafSrc &src, f3 o, f3 d, f3 inv, int negmask, float tmin, float tmax, HitRec &rec) {
    const unsigned *row = src.tab + 16 * negmask;
    unsigned pos[8];
    if (!ANY) {
#pragma unroll
        for (int w = 0; w < 8; ++w) pos[w] = row[8 + w];
    }
    unsigned pm = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
    // wave-uniform by construction; said explicitly, "4 g < n" below is a scalar compare -- left to itself the compiler keeps the seven
    // group conditions as lane masks in SGPR pairs, spills them into VGPR lanes and reads them back in every call (C2: 2.2 % of the kernel)
    const int n = __builtin_amdgcn_readfirstlane(src.nleaf);
#else
    const int n = src.nleaf;
#endif
    // the list is padded to a multiple of 4 (dummy boxes at positions >= n, masked off below); the boxes are read through the
    // constant address space with wave-uniform indices: scalar loads, SGPR operands, no LDS / vector-memory traffic
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) float *CBox;
#else
    typedef const float *CBox;
#endif
    const CBox cb = (CBox) (const void *) src.gbox;
    // closestHit's interval is open-ended (integrator.cpp:181: Interval(0.001, INF)): min(x, +inf) == x for the non-NaN x of a
    // regular ray, so the t.max operand is dropped -- the compiler cannot (fminf(NaN, inf) is inf)
    const bool openEnd = !ANY && tmax == __builtin_inff();
    const f3 ip = mk3(inv.x > 0.0f ? inv.x : 0.0f, inv.y > 0.0f ? inv.y : 0.0f, inv.z > 0.0f ? inv.z : 0.0f);
    const f3 in = mk3(inv.x > 0.0f ? 0.0f : inv.x, inv.y > 0.0f ? 0.0f : inv.y, inv.z > 0.0f ? 0.0f : inv.z);
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) {
        const int g = ANY ? 7 - gg : gg;
        if (4 * g < n) {                                                         // wave-uniform
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = ANY ? 3 - jj : jj;
                const int i = 4 * g + j;
                const float4 na = make_float4(cb[8 * i], cb[8 * i + 1], cb[8 * i + 2], cb[8 * i + 3]);
                const float4 nb = make_float4(cb[8 * i + 4], cb[8 * i + 5], 0.0f, 0.0f);
                const bool pass = openEnd ? slabRegularSel<true>(na, nb, o, ip, in, tmin, tmax) : slabRegularSel<false>(na, nb, o, ip, in, tmin, tmax);
                if (ANY) {
                    // anyHit's mask is in leaf order: the boxes are taken from the last to the first and every verdict is shifted in
                    // from the right as the carry of pm + pm (one v_addc_co_u32 instead of v_cndmask + shift + or)
#if defined(__HIP_DEVICE_COMPILE__)
                    const unsigned long long verdict = __ballot(pass);
                    unsigned long long carryOut;
                    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(pm), "=s"(carryOut) : "s"(verdict));
#else
                    pm = pm + pm + (pass ? 1u : 0u);
#endif
                    continue;
                }
                const unsigned at = (pos[i >> 2] >> (8 * (i & 3))) & 0xffu;
                pm |= (pass ? 1u : 0u) << at;
            }
        }
    }
    pm &= n >= 32 ? 0xffffffffu : (1u << n) - 1u;
    bool hitAnything = false, shrunk = false;
    int leafOff = 0, leafN = 0;
    while (true) {
        while (true) {
            if (leafN == 0 && pm != 0u) {
                const int k = __builtin_ctz(pm);
                pm &= pm - 1u;
                const int leaf = ANY ? k : (int) ((row[k >> 2] >> (8 * (k & 3))) & 0xffu);
                const float4 la = src.leafbox(leaf, 0), lb = src.leafbox(leaf, 1);
                bool pass = true;
                if (!ANY && shrunk) pass = slabRegularSel<false>(la, lb, o, ip, in, tmin, tmax);
                if (pass) { leafOff = __float_as_int(lb.z); leafN = __float_as_int(lb.w); }
            }
            const unsigned long long walking = __ballot(leafN == 0 && pm != 0u);
            if (walking == 0ull || __popcll(__ballot(leafN != 0)) >= JTX_LEAF_TRI_VOTE) break;
        }
        if (__ballot(leafN != 0) == 0ull) break;
        if (leafN != 0) {
            for (int i = 0; i < leafN; ++i) {
                const int prim = leafOff + i;
                float b1, b2, root;
                if (!triTest(src, prim, o, d, tmin, tmax, b1, b2, root)) continue;
                hitAnything = true;
                if (ANY) break;
                tmax = root; shrunk = true;
                rec.t = root; rec.prim = prim; rec.b1 = b1; rec.b2 = b2;
            }
            if (ANY && hitAnything) pm = 0u;
            leafN = 0;
        }
    }
    return hitAnything;
}

// rays the leaf list may take: regular, with |o| <= 2^60 -- the list's planes are bounded the same way (buildLeafTables, jtx_capi.hip), so
// lo - o and hi - o are finite and the vanishing term of slabRegularSel's fma IS a zero (inf * 0 would be a NaN where AABB::hit has +-inf)
constexpr float LEAF_RANGE = 1152921504606846976.0f;   // 2^60
JD bool leafRayOk(f3 o, f3 inv, float tmin, float tmax) {
    return finiteNonZero(inv.x) && finiteNonZero(inv.y) && finiteNonZero(inv.z) &&
           fabsf(o.x) <= LEAF_RANGE && fabsf(o.y) <= LEAF_RANGE && fabsf(o.z) <= LEAF_RANGE && tmin == tmin && tmax == tmax;
}

template <bool ANY, bool COUNT>
JD bool traverseNoStack(const LeafSrc &src, int num_nodes, f3 o, f3 d, float tmin, float tmax, HitRec &rec, Counters9 &cnt) {
    static_assert(!COUNT, "the counted kernels reproduce the reference's node visits: binary records only");
    if (num_nodes == 0) return false;
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const int negmask = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
    if (__builtin_expect(__ballot(!leafRayOk(o, inv, tmin, tmax)) == 0ull, 1)) {
        // Scene::closestHit tests the ROOT's box first (scene.cpp:20-24) and a ray that fails it has no hit; the leaf list never looks
        // at the root (leaf boxes nest in it and the slab test is monotone in the planes).  A wave that holds nothing but camera rays
        // -- every wave of the frame's surround, 44 % of C2's paths -- asks, and skips phase A when all of them miss.  The root
        // record is read with wave-uniform addresses: scalar loads, SGPR operands.
        if (!ANY && src.groot && __ballot(!src.fresh) == 0ull) {
#if defined(__HIP_DEVICE_COMPILE__)
            typedef const __attribute__((address_space(4))) float *CBox;
#else
            typedef const float *CBox;
#endif
            const CBox cb = (CBox) (const void *) src.groot;
            const float4 na = make_float4(cb[0], cb[1], cb[2], cb[3]);
            const float4 nb = make_float4(cb[4], cb[5], 0.0f, 0.0f);
            if (__ballot(slabRegular(na, nb, o, inv, tmin, tmax)) == 0ull) return false;
        }
        return traverseLeaves<ANY>(src, o, d, inv, negmask, tmin, tmax, rec);
    }
    return traverseThreaded<ANY, false, false>(src, num_nodes, o, d, inv, negmask, tmin, tmax, rec, cnt);
}

struct Surface { f3 point, normal; f2 uv; int material; };


// The accept branch of Mesh::tClosestHit (mesh.hpp:129-145) + setFaceNormal (material.hpp:36-39)
JD Surface makeSurface(const float4 *shade, const HitRec &h, f3 o, f3 d) {
    const float4 s0 = shade[4 * h.prim + 0], s1 = shade[4 * h.prim + 1], s2 = shade[4 * h.prim + 2], s3 = shade[4 * h.prim + 3];
    const f3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
    Surface r;
    r.point = o + h.t * d;
    const float b0 = (1.0f - h.b1 - h.b2);
    const f3 n = b0 * n0 + h.b1 * n1 + h.b2 * n2;
    r.normal = dot(d, n) < 0.0f ? n : -n;
    r.uv = mk2(s2.y * b0 + s2.w * h.b1 + s3.y * h.b2, s2.z * b0 + s3.x * h.b1 + s3.z * h.b2);
    r.material = __float_as_int(s3.w);
    return r;
}

// Light::sample (lights.hpp:36-54)
struct LightSample { f3 p, radiance, wi; float pdf; };
JD bool lightSample(const DLight &l, f3 p, LightSample &ls) {
    const f3 pos = a3(l.position), I = a3(l.intensity);
    if (l.type == 0) {
        ls.p = pos;
        ls.wi = normalize(pos - p);
        ls.radiance = l.scale * I / lenSqr(pos - p);
        ls.pdf = 1.0f;
        return true;
    }
    if (l.type == 1) {
        ls.p = p - pos * 2.0f * l.scene_radius;
        ls.wi = -pos;
        ls.radiance = l.scale * I;
        ls.pdf = 1.0f;
        return true;
    }
    return false;
}

JD float powerHeuristic(float nf, float fPdf, float ng, float gPdf) {   // integrator.cpp:6-10
    const float f = nf * fPdf, g = ng * gPdf;
    return f * f / (f * f + g * g);
}

// Camera state derived on the host by Camera::init (camera.cpp:7-31)
struct DCam {
    float center[3], vp00[3], du[3], dv[3], defocus_u[3], defocus_v[3];
    float defocus_angle;
    int xs, ys;
};

// Camera::getRay (camera.hpp:127-139), called as getRay(col,row,stratum) (camera.cpp:103)
JD void cameraRay(const DCam &k, uint32_t col, uint32_t row, uint32_t stratum, Rng &rng, f3 &o, f3 &d) {
    const uint32_t x = stratum % (uint32_t) k.xs, y = stratum / (uint32_t) k.xs;
    const float dx = rng.f(), dy = rng.f();
    const float offx = ((float) x + dx) / (float) k.xs, offy = ((float) y + dy) / (float) k.ys;
    const f3 sample = a3(k.vp00) + ((float) col + offx) * a3(k.du) + ((float) row + offy) * a3(k.dv);
    f3 origin = a3(k.center);
    if (!(k.defocus_angle <= 0.0f)) {
        float px, py;
        while (true) {                                       // RNG::sampleUnitDisc rand.hpp:179-184
            px = -1.0f + (1.0f - -1.0f) * rng.f();
            py = -1.0f + (1.0f - -1.0f) * rng.f();
            if (px * px + py * py + 0.0f * 0.0f < 1.0f) break;
        }
        origin = a3(k.center) + (px * a3(k.defocus_u)) + (py * a3(k.defocus_v));
    }
    (void) rng.f();                                          // ray time
    o = origin; d = sample - origin;
}

} // namespace jtx
