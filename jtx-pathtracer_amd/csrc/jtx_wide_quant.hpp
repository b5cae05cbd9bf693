// jtx_wide_quant.hpp -- outward quantisation of the child boxes of an 8-ary node (host builder AND device refit).
//
// A wide node stores its children's boxes on a grid of its own: plane = origin + q * 2^e per axis, q in 0..255, origin =
// the node's min corner, e the smallest exponent with origin + 255 * 2^e >= max corner.  Child planes are rounded OUTWARD
// (q_lo = the largest grid plane <= the child's min, q_hi = the smallest >= its max), decided in EXACT arithmetic
// (gridCmp: an error-free two-sum in double), so the quantised box provably contains the exact one -- the premise of the
// traversal's equivalence argument.  Shared between jtx_capi.hip's WideBuilder (host) and
// jtx_refit.hip (device): the same code decides both.
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

// sign of (p + q * cell) - x in exact arithmetic (q * cell is exact in double; two-sum for the addition)
JTXQ_HD int gridCmp(float p, int q, float cell, float x) {
    const double a = (double) p, b = (double) q * (double) cell;
    const double t = a + b, bb = t - a, err = (a - (t - bb)) + (b - bb);
    if (t != (double) x) return t < (double) x ? -1 : 1;
    return err < 0 ? -1 : (err > 0 ? 1 : 0);
}
JTXQ_HD bool finite3(const float v[3]) { return isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]); }

// grid of a node: exponent byte (e + 127) and cell per axis.  false: the node cannot carry a grid (the scene then keeps
// the binary records only)
JTXQ_HD bool nodeGrid(const float pmin[3], const float pmax[3], uint32_t ebyte[3], float cell[3]) {
    if (!finite3(pmin) || !finite3(pmax)) return false;
    for (int k = 0; k < 3; ++k) {
        const double ext = (double) pmax[k] - (double) pmin[k];
        int e = ext > 0 ? ilogb(ext / 255.0) : kWideMinExp;
        if (e < kWideMinExp) e = kWideMinExp;              // cell / d must stay a normal float (exact scaling)
        while (e <= kWideMaxExp && gridCmp(pmin[k], 255, ldexpf(1.0f, e), pmax[k]) < 0) ++e;
        if (e > kWideMaxExp || fabsf(pmin[k]) > kWideCoordMax || fabsf(pmax[k]) > kWideCoordMax) return false;
        ebyte[k] = (uint32_t) (e + 127); cell[k] = ldexpf(1.0f, e);
    }
    return true;
}

// one child box on the node's grid, rounded outward.  false: the child does not nest in the node (or is inverted)
JTXQ_HD bool quantiseChild(const float pmin[3], const float pmax[3], const float cell[3], const float cmin[3], const float cmax[3],
                           uint8_t qlo[3], uint8_t qhi[3]) {
    for (int k = 0; k < 3; ++k) {
        if (!(cmin[k] >= pmin[k] && cmax[k] <= pmax[k] && cmin[k] <= cmax[k])) return false;   // nesting is the premise
        const float p = pmin[k], sc = cell[k];
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

} // namespace jtxq
