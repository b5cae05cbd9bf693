// jtx_refit.hip -- device refit after a transform edit (Display's translate / scale edits -> recalculateTransform ->
// rebuildBVH_, display.cpp:545-588, Scene::rebuildBVH scene.hpp:60-64).
//
// The reference rebuilds the whole BVH on the host after every edit.  Here the edited scene stays on the device: with the
// TOPOLOGY of the last build kept, everything that depends on vertex positions is recomputed in five small kernels --
//   k_refit_prims    : Mesh::transform applied to the captured object-space vertices / normals (the fp32 operations of
//                      Mesh::getVertices / getNormals, mesh.hpp:71-77,92-97) -> triangle records, shading normals, primitive boxes
//   k_refit_leaves   : leaf boxes = union of their primitives' boxes (tBounds, mesh.hpp:79-84)
//   k_refit_level    : interior boxes = union of the two children's, deepest level first (initBranch, bvh.hpp:31-37)
//   k_refit_threaded : the boxes into the 8 direction-sign orderings of the stackless records
//   k_refit_wide     : every 8-ary node re-gridded and its children re-quantised OUTWARD with the builder's own exact
//                      arithmetic (jtx_wide_quant.hpp), leaf records get their exact boxes
// min / max are exact, so every box is bit for bit the box a fresh build computes for the same set of primitives.  What a
// refit does NOT reproduce is the reference's choice of topology for the edited geometry: the frame is a correct render of
// the edited scene -- for scenes built with maxPrimsInNode = 1 every regular ray finds the same hit distance, and the same
// primitive except among exactly equal distances -- but it is flagged (scene_info.refitted) and is not
// claimed bit-identical to the frame after Scene::rebuildBVH; for that, create the scene again.
#include "jtx_scene_dev.hpp"
#include "jtx_launch.hpp"
#include "jtx_wide_quant.hpp"

namespace jtx {

JD void xformPoint(const float *m, f3 v, f3 &o) {            // Transform::applyToPoint, row by row (jtx_bvh_build.cpp xformPoint)
    o.x = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
    o.y = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7];
    o.z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11];
}
JD void xformNormal(const float *m, f3 v, f3 &o) {           // Transform::applyToNormal: upper 3x3 (jtx_capi.hip xformNormal)
    o.x = m[0] * v.x + m[1] * v.y + m[2] * v.z;
    o.y = m[4] * v.x + m[5] * v.y + m[6] * v.z;
    o.z = m[8] * v.x + m[9] * v.y + m[10] * v.z;
}

__global__ void __launch_bounds__(256) k_refit_prims(RefitArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.num_prims) return;
    const float4 s0 = a.prim_src[5 * (size_t) i], s1 = a.prim_src[5 * (size_t) i + 1], s2 = a.prim_src[5 * (size_t) i + 2],
                 s3 = a.prim_src[5 * (size_t) i + 3], s4 = a.prim_src[5 * (size_t) i + 4];
    const float *m = a.mesh_xf + 16 * (size_t) __float_as_int(s4.z);
    f3 v0, v1, v2, n0, n1, n2;
    xformPoint(m, mk3(s0.x, s0.y, s0.z), v0); xformPoint(m, mk3(s0.w, s1.x, s1.y), v1); xformPoint(m, mk3(s1.z, s1.w, s2.x), v2);
    xformNormal(m, mk3(s2.y, s2.z, s2.w), n0); xformNormal(m, mk3(s3.x, s3.y, s3.z), n1); xformNormal(m, mk3(s3.w, s4.x, s4.y), n2);
    const f3 e1 = v1 - v0, e2 = v2 - v0;                      // v0v1, v0v2 (mesh.hpp:109-110)
    float4 *t = a.tris + 3 * (size_t) i;
    const float keepType = t[2].y;
    t[0] = make_float4(v0.x, v0.y, v0.z, e1.x);
    t[1] = make_float4(e1.y, e1.z, e2.x, e2.y);
    t[2] = make_float4(e2.z, keepType, 0.0f, 0.0f);
    float4 *sh = a.shade + 4 * (size_t) i;
    const float4 k2 = sh[2];
    sh[0] = make_float4(n0.x, n0.y, n0.z, n1.x);
    sh[1] = make_float4(n1.y, n1.z, n2.x, n2.y);
    sh[2] = make_float4(n2.z, k2.y, k2.z, k2.w);             // uvs and the material id stay
    a.pbox[2 * (size_t) i] = make_float4(fminf(fminf(v0.x, v1.x), v2.x), fminf(fminf(v0.y, v1.y), v2.y), fminf(fminf(v0.z, v1.z), v2.z), 0.0f);
    a.pbox[2 * (size_t) i + 1] = make_float4(fmaxf(fmaxf(v0.x, v1.x), v2.x), fmaxf(fmaxf(v0.y, v1.y), v2.y), fmaxf(fmaxf(v0.z, v1.z), v2.z), 0.0f);
}

__global__ void __launch_bounds__(256) k_refit_leaves(RefitArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.num_leaves) return;
    const int node = a.leaf_nodes[i];
    const float4 nb = a.nbox[2 * (size_t) node + 1];
    const int off = __float_as_int(nb.z), n = __float_as_int(nb.w);
    float4 lo = a.pbox[2 * (size_t) off], hi = a.pbox[2 * (size_t) off + 1];
    for (int k = 1; k < n; ++k) {
        const float4 l2 = a.pbox[2 * (size_t) (off + k)], h2 = a.pbox[2 * (size_t) (off + k) + 1];
        lo.x = fminf(lo.x, l2.x); lo.y = fminf(lo.y, l2.y); lo.z = fminf(lo.z, l2.z);
        hi.x = fmaxf(hi.x, h2.x); hi.y = fmaxf(hi.y, h2.y); hi.z = fmaxf(hi.z, h2.z);
    }
    a.nbox[2 * (size_t) node] = make_float4(lo.x, hi.x, lo.y, hi.y);
    a.nbox[2 * (size_t) node + 1] = make_float4(lo.z, hi.z, nb.z, nb.w);
}

__global__ void __launch_bounds__(256) k_refit_level(RefitArgs a, int begin, int count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const int node = a.level_nodes[begin + i];
    const float4 nb = a.nbox[2 * (size_t) node + 1];
    const int second = __float_as_int(nb.z);
    const float4 a0 = a.nbox[2 * (size_t) (node + 1)], a1 = a.nbox[2 * (size_t) (node + 1) + 1];
    const float4 b0 = a.nbox[2 * (size_t) second], b1 = a.nbox[2 * (size_t) second + 1];
    a.nbox[2 * (size_t) node] = make_float4(fminf(a0.x, b0.x), fmaxf(a0.y, b0.y), fminf(a0.z, b0.z), fmaxf(a0.w, b0.w));
    a.nbox[2 * (size_t) node + 1] = make_float4(fminf(a1.x, b1.x), fmaxf(a1.y, b1.y), nb.z, nb.w);
}

__global__ void __launch_bounds__(256) k_refit_threaded(RefitArgs a) {
    const size_t r = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (r >= (size_t) 8 * a.num_nodes) return;
    const int node = a.rec_node[r];
    const float4 b0 = a.nbox[2 * (size_t) node], b1 = a.nbox[2 * (size_t) node + 1];
    a.tnodes[2 * r] = b0;
    const float4 keep = a.tnodes[2 * r + 1];
    a.tnodes[2 * r + 1] = make_float4(b1.x, b1.y, keep.z, keep.w);      // link / leaf words stay
}

// Scene creation: the 8 x num_nodes stackless records (layout: traverseThreaded in jtx_scene_dev.hpp) written on the device
// from the binary nodes, every node's position in each octant's near-first order (host: one top-down pass per octant)
// and the subtree sizes -- the host used to fill and upload the 134 MB array itself (atrium: 16 of 67 ms).
__global__ void __launch_bounds__(256) k_build_threaded(const float4 *nbox, const int *pos, const int *size, int nn, float4 *tnodes, int *rec_node) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= nn) return;
    const float4 b0 = nbox[2 * (size_t) g], b1 = nbox[2 * (size_t) g + 1];
    const int offset = __float_as_int(b1.z), nprims = __float_as_int(b1.w);
    const int sz = size[g];
    for (int k = 0; k < 8; ++k) {
        const int i = pos[(size_t) k * nn + g];
        int z, w;
        if (nprims == 0) {                                   // interior: where the walk goes on when the box is missed
            const int behind = i + sz;
            z = behind < nn ? k * nn + behind : -1;
            w = 0;
        } else {                                             // leaf: its primitives; bit 31 marks the last record of the order
            z = offset;
            w = nprims | (i + 1 == nn ? (int) 0x80000000u : 0);
        }
        const size_t r = (size_t) k * nn + i;
        rec_node[r] = g;
        tnodes[2 * r] = b0;
        tnodes[2 * r + 1] = make_float4(b1.x, b1.y, __int_as_float(z), __int_as_float(w));
    }
}

JD void nodeCorners(const RefitArgs &a, int node, float lo[3], float hi[3]) {
    const float4 b0 = a.nbox[2 * (size_t) node], b1 = a.nbox[2 * (size_t) node + 1];
    lo[0] = b0.x; hi[0] = b0.y; lo[1] = b0.z; hi[1] = b0.w; lo[2] = b1.x; hi[2] = b1.y;
}

__global__ void __launch_bounds__(128) k_refit_wide(RefitArgs a) {
    const int w = blockIdx.x * 128 + threadIdx.x;
    if (w >= a.num_wide) return;
    const int *rec = a.wide_map + 16 * (size_t) w;
    const int at = rec[0], b = rec[1], ni = rec[2], nl = rec[3], leafBase = rec[4];
    float pmin[3], pmax[3];
    nodeCorners(a, b, pmin, pmax);
    jtxq::NodeGrid grid;
    if (!jtxq::nodeGrid(pmin, pmax, grid)) { atomicExch(a.wide_fail, 1); return; }
    uint8_t qlo[3][8] = {}, qhi[3][8] = {};
    for (int s = 0; s < ni + nl; ++s) {
        float cmin[3], cmax[3];
        nodeCorners(a, rec[8 + s], cmin, cmax);
        uint8_t lo3[3], hi3[3];
        if (!jtxq::quantiseChild(grid, pmin, pmax, cmin, cmax, lo3, hi3)) { atomicExch(a.wide_fail, 1); return; }
        for (int k = 0; k < 3; ++k) { qlo[k][s] = lo3[k]; qhi[k][s] = hi3[k]; }
        if (s >= ni) {                                        // leaf record: the exact box (offset / count stay)
            uint4 *lr = a.wide + leafBase + 2 * (s - ni);
            const uint4 keep = lr[1];
            lr[0] = make_uint4(__float_as_uint(cmin[0]), __float_as_uint(cmax[0]), __float_as_uint(cmin[1]), __float_as_uint(cmax[1]));
            lr[1] = make_uint4(__float_as_uint(cmin[2]), __float_as_uint(cmax[2]), keep.z, keep.w);
        }
        if (at == (int) jtxq::kRootNode) {                    // the root-peel record carries the exact boxes of the root's children
            uint32_t *peel = (uint32_t *) (a.wide + jtxq::kPeelRec);
            jtxq::encodePeelBox(peel, s, cmin, cmax);
        }
    }
    uint32_t nd[16];
    uint4 *n = a.wide + at;
    jtxq::encodeGridAndPlanes(nd, grid, ni, ni + nl, qlo, qhi);
    for (int g = 0; g < 4; ++g) n[g] = make_uint4(nd[4 * g], nd[4 * g + 1], nd[4 * g + 2], nd[4 * g + 3]);
    // the tail granule (children base, visiting orders) and the head of the root-peel record depend on the topology only
}

} // namespace jtx

using namespace jtx;

// level_begin: num_levels + 1 offsets into level_nodes, deepest interior level LAST in the array = processed first
hipError_t jtx_launch_refit(const RefitArgs &a, const int *level_begin, int num_levels, hipStream_t st) {
    auto blocks = [](size_t n, int b) { return dim3((unsigned) ((n + b - 1) / b)); };
    if (a.num_prims) hipLaunchKernelGGL(k_refit_prims, blocks(a.num_prims, 256), dim3(256), 0, st, a);
    if (a.num_leaves) hipLaunchKernelGGL(k_refit_leaves, blocks(a.num_leaves, 256), dim3(256), 0, st, a);
    for (int l = num_levels - 1; l >= 0; --l) {
        const int count = level_begin[l + 1] - level_begin[l];
        if (count > 0) hipLaunchKernelGGL(k_refit_level, blocks(count, 256), dim3(256), 0, st, a, level_begin[l], count);
    }
    if (a.num_nodes) hipLaunchKernelGGL(k_refit_threaded, blocks((size_t) 8 * a.num_nodes, 256), dim3(256), 0, st, a);
    if (a.num_wide && a.wide) hipLaunchKernelGGL(k_refit_wide, blocks(a.num_wide, 128), dim3(128), 0, st, a);
    return hipGetLastError();
}

hipError_t jtx_launch_refit_prims(const RefitArgs &a, hipStream_t st) {
    if (a.num_prims) hipLaunchKernelGGL(k_refit_prims, dim3((unsigned) ((a.num_prims + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t jtx_launch_build_threaded(const float4 *nbox, const int *pos, const int *size, int nn, float4 *tnodes, int *rec_node, hipStream_t st) {
    if (nn <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_build_threaded, dim3((unsigned) ((nn + 255) / 256)), dim3(256), 0, st, nbox, pos, size, nn, tnodes, rec_node);
    return hipGetLastError();
}
