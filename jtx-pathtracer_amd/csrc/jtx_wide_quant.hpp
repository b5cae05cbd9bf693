// jtx_wide_quant.hpp -- the 8-ary node set of traverseWide: its layout in memory, the outward quantisation of child boxes and the
// encoders, shared by the host builder (jtx_capi.hip), the device builder (jtx_build_dev.hip) and the device refit (jtx_refit.hip):
// ONE piece of code decides every bit of a node, whoever writes it.
//
// Layout, round 4 (16-byte granules):
//   root-peel record   granules 0..13 (kPeelRec): the root's children with their EXACT boxes, what a first step through scalar loads reads
//                      (measured and dropped: tools/experiments/r04_slip_rootpeel_tails.patch; the record stays in the layout, tests walk it):
//                      [group word | the visiting orders of octants 0-3][#children | orders of octants 4-7][8 children x exact box, 6 floats]
//   root node          granules 16.. (kRootNode): the root as an ordinary node
//   children blocks    from granule 32 (kFirstBlock): for a node with ni interior children and nl leaves
//                          [ni x node, kNodeG granules each][nl x leaf record, 2 granules]
//   node (192 B)       g0 [origin.xyz (float: the node's min corner) | ex ey ez, ni << 24, (ni + nl) << 28]   plane = origin + q 2^e
//                      g1 [lo.x x8 | lo.y x8]  g2 [lo.z x8 | hi.x x8]  g3 [hi.y x8 | hi.z x8]   (8-bit planes, slot s = byte s)
//                      g4..g11  ONE TAIL PER DIRECTION-SIGN OCTANT q [children block granule | the octant's 24-bit visiting order (3 bits
//                          per position) | one-hot position of slots 0-3, a byte each | of slots 4-7]: a ray reads the tail of its
//                          octant -- base, order and the words that turn its eight pass / miss bytes into the pending mask of the
//                          order list (wideNodePend: the permutation of the hit bits costs one AND and one byte sum, not 26 instructions)
//   leaf record (32 B) the exact leaf box + primitivesOffset + numPrimitives  [min.x max.x min.y max.y][min.z max.z offset count]
// Rounds 1-3 had TWO tails (octants 0-3 / 4-7: base + four 24-bit orders; 96-byte nodes) and permuted
// the hit mask in registers.  With the per-octant tails C3 320 -> 305 ms, C5 303 -> 287 (SQ_INSTS_VALU -6.5 %, L2 requests +7 % from the
// larger nodes); see DESIGN.md section 10.  Other layouts built, taken through the whole parity suite and measured in round 4
// (profiles/r04_wide_layouts.md), each behind the encoders of this file:
//   * the children base packed INTO the header beside a 24-bit integer origin (plane = (k + q) 2^e), so that an anyHit step reads four
//     granules instead of five: (a) 64-byte nodes on 64-byte boundaries, order granules at the end of the block: C3 339 ms against 319;
//     (b) order granule behind its node: 331 ms.  3.4 % fewer load instructions, but unpacking the header costs 12-16 VALU
//     instructions per node step (+4.0 % SQ_INSTS_VALU), and the kernel is short of issue slots first;
//   * ONE tail granule (80-byte nodes), octant q >= 4 walking the order of octant 7 - q backwards (all three signs
//     flipped = every near / far decision of the treelet flipped): same instruction count in the node loop, C3 326 / C5 317 ms
//     against 321 / 303 -- the per-ray direction flag and class cost two more live registers in kernels that spill 94-161;
//   * root peel: the root's children on their exact boxes through scalar loads: C3 327 (+2 %),
//     C5 297 (-2 %); the 48 box words take 10 more spilled SGPRs and ~85 v_readlane per bounce, and the five vector loads it
//     removes were ONE coalesced request per wave each (all 64 lanes read the root), not 64;
//   * children blocks padded to 64 / 128 bytes on top of the per-octant tails: +-0.
//
// Quantisation: a node's grid is plane = origin + q * 2^e per axis, q in 0..255, origin = the node's min corner, e the smallest
// exponent with origin + 255 * 2^e >= max corner.  Child planes are rounded OUTWARD (q_lo = the largest grid plane <= the child's min,
// q_hi = the smallest >= its max), decided in EXACT arithmetic (gridCmp: an error-free two-sum in double), so the quantised box
// provably contains the exact one -- the premise of the traversal's equivalence argument.
#pragma once
#include <math.h>
#include <stdint.h>
#if defined(__HIPCC__)
#define JTXQ_HD __host__ __device__ inline
#else
#define JTXQ_HD inline
#endif

namespace jtxq {

constexpr int kWideMinExp = -60, kWideMaxExp = 40;   // cell = 2^e; with |1/d| in [2^-40, 2^40] (WIDE_RANGE) cell / d is exact
constexpr float kWideCoordMax = 1099511627776.0f;    // 2^40

constexpr uint32_t kTails = 8;                       // one tail granule PER OCTANT [children base | its 24-bit order | the one-hot POSITION of slots 0-3 | of slots 4-7]
constexpr uint32_t kNodeG = 4 + kTails;              // granules of a node record
constexpr uint32_t kPeelRec = 0, kPeelBoxes = 2, kRootNode = 16, kFirstBlock = 32;
constexpr uint32_t kMaxGranules = 1u << 28;          // the children base shares its word with a 4-bit count in the kernel's group state
JTXQ_HD uint32_t blockGranules(int ni, int nl) { return kNodeG * (uint32_t) ni + 2u * (uint32_t) nl; }
JTXQ_HD uint32_t nodeAt(uint32_t base, int s) { return base + kNodeG * (uint32_t) s; }
JTXQ_HD uint32_t leafAt(uint32_t base, int ni, int l) { return base + kNodeG * (uint32_t) ni + 2u * (uint32_t) l; }
JTXQ_HD uint32_t groupWord(uint32_t base, int ni) { return base | (uint32_t) ni << 28; }   // the kernel's per-group state: where the children stand

// sign of (p + q * cell) - x in exact arithmetic (q * cell is exact in double; two-sum for the addition)
JTXQ_HD int gridCmp(float p, int q, float cell, float x) {
    const double a = (double) p, b = (double) q * (double) cell;
    const double t = a + b, bb = t - a, err = (a - (t - bb)) + (b - bb);
    if (t != (double) x) return t < (double) x ? -1 : 1;
    return err < 0 ? -1 : (err > 0 ? 1 : 0);
}
JTXQ_HD bool finite3(const float v[3]) { return isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]); }

struct NodeGrid { uint32_t ebyte[3]; float cell[3], origin[3]; };

// grid of a node: exponent byte (e + 127) and cell per axis.  false: the node cannot carry a grid (the scene then keeps
// the binary records only)
JTXQ_HD bool nodeGrid(const float pmin[3], const float pmax[3], NodeGrid &g) {
    if (!finite3(pmin) || !finite3(pmax)) return false;
    for (int k = 0; k < 3; ++k) {
        const double ext = (double) pmax[k] - (double) pmin[k];
        int e = ext > 0 ? ilogb(ext / 255.0) : kWideMinExp;
        if (e < kWideMinExp) e = kWideMinExp;              // cell / d must stay a normal float (exact scaling)
        while (e <= kWideMaxExp && gridCmp(pmin[k], 255, ldexpf(1.0f, e), pmax[k]) < 0) ++e;
        if (e > kWideMaxExp || fabsf(pmin[k]) > kWideCoordMax || fabsf(pmax[k]) > kWideCoordMax) return false;
        g.ebyte[k] = (uint32_t) (e + 127); g.cell[k] = ldexpf(1.0f, e); g.origin[k] = pmin[k];
    }
    return true;
}

// one child box on the node's grid, rounded outward.  false: the child does not nest in the node (or is inverted)
JTXQ_HD bool quantiseChild(const NodeGrid &g, const float pmin[3], const float pmax[3], const float cmin[3], const float cmax[3],
                           uint8_t qlo[3], uint8_t qhi[3]) {
    for (int k = 0; k < 3; ++k) {
        if (!(cmin[k] >= pmin[k] && cmax[k] <= pmax[k] && cmin[k] <= cmax[k])) return false;   // nesting is the premise
        const float p = g.origin[k], sc = g.cell[k];
        int q = (int) floor(((double) cmin[k] - (double) p) / (double) sc);
        q = q < 0 ? 0 : (q > 255 ? 255 : q);
        while (q > 0 && gridCmp(p, q, sc, cmin[k]) > 0) --q;
        while (q < 255 && gridCmp(p, q + 1, sc, cmin[k]) <= 0) ++q;
        if (gridCmp(p, q, sc, cmin[k]) > 0) return false;
        qlo[k] = (uint8_t) q;
        q = (int) ceil(((double) cmax[k] - (double) p) / (double) sc);
        q = q < 0 ? 0 : (q > 255 ? 255 : q);
        while (q < 255 && gridCmp(p, q, sc, cmax[k]) < 0) ++q;
        while (q > 0 && gridCmp(p, q - 1, sc, cmax[k]) >= 0) --q;
        if (gridCmp(p, q, sc, cmax[k]) < 0) return false;
        qhi[k] = (uint8_t) q;
    }
    return true;
}

JTXQ_HD uint32_t pack4(const uint8_t *q) { return (uint32_t) q[0] | (uint32_t) q[1] << 8 | (uint32_t) q[2] << 16 | (uint32_t) q[3] << 24; }

// the node's first four granules: nd[0..15]
JTXQ_HD void encodeGridAndPlanes(uint32_t *nd, const NodeGrid &g, int ni, int nchild, const uint8_t qlo[3][8], const uint8_t qhi[3][8]) {
    union { float f; uint32_t u; } c;
    for (int a = 0; a < 3; ++a) { c.f = g.origin[a]; nd[a] = c.u; }
    nd[3] = g.ebyte[0] | g.ebyte[1] << 8 | g.ebyte[2] << 16 | (uint32_t) ni << 24 | (uint32_t) nchild << 28;
    nd[4] = pack4(qlo[0]); nd[5] = pack4(qlo[0] + 4); nd[6] = pack4(qlo[1]); nd[7] = pack4(qlo[1] + 4);
    nd[8] = pack4(qlo[2]); nd[9] = pack4(qlo[2] + 4); nd[10] = pack4(qhi[0]); nd[11] = pack4(qhi[0] + 4);
    nd[12] = pack4(qhi[1]); nd[13] = pack4(qhi[1] + 4); nd[14] = pack4(qhi[2]); nd[15] = pack4(qhi[2] + 4);
}

// the node's tail granule from the children base and the visiting orders of the 8 octants (3 bits per position, first visited
// first).  false: octant 7 - q is not the reverse of octant q (cannot happen for orders made by the near-first rule)
JTXQ_HD bool encodeTail(uint32_t *w, uint32_t base, const uint32_t perm[8], int nchild) {      // w: 4 * kTails words
    for (int q = 0; q < 4; ++q)
        for (int p = 0; p < nchild; ++p)
            if (((perm[q] >> (3 * p)) & 7u) != ((perm[7 - q] >> (3 * (nchild - 1 - p))) & 7u)) return false;
    // byte s of the two one-hot words = 1 << (position of slot s in this octant's order), 0 for a slot without a child: a per-child
    // pass / miss byte mask ANDed with them and summed over its bytes IS the pending mask in position space (wideNodePend)
    for (uint32_t t = 0; t < kTails; ++t) {
        uint32_t oh[2] = {0u, 0u};
        for (int p = 0; p < nchild; ++p) { const uint32_t sl = (perm[t] >> (3 * p)) & 7u; oh[sl >> 2] |= (1u << p) << (8 * (sl & 3u)); }
        w[4 * t + 0] = base; w[4 * t + 1] = perm[t] & 0x00ffffffu; w[4 * t + 2] = oh[0]; w[4 * t + 3] = oh[1];
    }
    return true;
}

// the root-peel record (14 granules = 56 words): group word + the orders, #children, then the EXACT boxes of the root's children in slot order
// [min.x max.x min.y max.y min.z max.z]; unused slots zero (masked by nchild in the kernel)
JTXQ_HD void encodePeelHeader(uint32_t *rec, uint32_t base, int ni, int nchild, const uint32_t perm[8]) {
    rec[0] = groupWord(base, ni); rec[4] = (uint32_t) nchild; rec[5] = rec[6] = rec[7] = 0u;
    for (uint32_t t = 0; t < 2u; ++t) {                                             // four 24-bit orders back to back; octants 4..7 behind #children
        const uint32_t *pm = perm + 4 * t;
        rec[4 * t + 1] = pm[0] | (pm[1] & 0xffu) << 24; rec[4 * t + 2] = pm[1] >> 8 | (pm[2] & 0xffffu) << 16; rec[4 * t + 3] = pm[2] >> 16 | pm[3] << 8;
    }
}
JTXQ_HD void encodePeelBox(uint32_t *rec, int slot, const float cmin[3], const float cmax[3]) {
    union { float f; uint32_t u; } c;
    uint32_t *b = rec + 4 * kPeelBoxes + 6 * slot;
    for (int a = 0; a < 3; ++a) { c.f = cmin[a]; b[2 * a] = c.u; c.f = cmax[a]; b[2 * a + 1] = c.u; }
}

} // namespace jtxq
