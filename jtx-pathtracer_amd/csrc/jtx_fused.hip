// jtx_fused.hip -- pixel-persistent integrator, fused form (integrator 4 while it is being evaluated).
//
// Like k_render_pixels a lane owns a pixel and keeps its path in registers, but one round of the path
// loop is restructured so that every lane has TWO rays to trace back to back -- the shadow ray of the
// vertex just shaded and the extension ray to the next vertex -- and traces them inside one
// traversal loop without reconverging in between.  The wave then waits for max over lanes of
// (len_shadow + len_extension) instead of max(len_extension) + max(len_shadow), and the shading code
// runs once per round instead of twice.  The light-sample contribution is computed before its shadow
// ray is traced and added afterwards if the ray got through -- the same sum, in the same order, as
// integrator.cpp:194-196.  Output is bit-identical to the other integrators.
#include "jtx_scene_dev.hpp"
#include "jtx_launch.hpp"

namespace jtx {

constexpr int FBLOCK = 256;
#ifndef JTX_SWITCH_VOTE
#define JTX_SWITCH_VOTE 12      // lanes done with their first ray that end the interior phase (to switch rays)
#endif

JD unsigned char fToByte(float v) {                                        // image.hpp:9-16,47-52
    const float g = v > 0.0f ? sqrtf(v) : 0.0f;
    const float c = clampf(g, 0.0f, 0.999f);
    return (unsigned char) (int) (255.999f * c);
}

template <bool COUNT, bool LDS_SCENE, int MASK>
__global__ void __launch_bounds__(FBLOCK, 8) k_render_fused(RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const DevScene &sc = p.scene;
    float4 *lds_tnodes = (float4 *) smem;                       // [8 threaded node orderings][tris], stackless
    float4 *lds_tris = lds_tnodes + 2 * 8 * sc.num_nodes;
    if (LDS_SCENE) {
        const int nn = 2 * 8 * sc.num_nodes, nt = 3 * sc.num_prims;
        for (int i = threadIdx.x; i < nn; i += FBLOCK) lds_tnodes[i] = sc.tnodes[i];
        for (int i = threadIdx.x; i < nt; i += FBLOCK) lds_tris[i] = sc.tris[i];
        __syncthreads();
    }
    GlobalSrc src; src.tnodes = LDS_SCENE ? lds_tnodes : sc.tnodes; src.tris = LDS_SCENE ? lds_tris : sc.tris;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int owned = blockIdx.x >> 2;
    const int tile = p.tile_rank + owned * p.tile_world;
    const int sub = ((blockIdx.x & 3) << 2) | wave;
    const int trow = tile / p.tiles_x, tcol = tile - trow * p.tiles_x;
    const int row = trow * 32 + (sub >> 2) * 8 + (lane >> 3);
    const int col = tcol * 32 + (sub & 3) * 8 + (lane & 7);
    const bool inside = row < p.height && col < p.width;
    const size_t pix = inside ? (size_t) row * p.width + col : 0;

    Counters9 cnt = {};
    ShadeCtx ctx; ctx.materials = sc.materials; ctx.textures = sc.textures; ctx.texels = sc.texels;

    f3 acc = mk3(0.0f);
    if (inside && p.sample_begin > 0) acc = mk3(p.acc[3 * pix], p.acc[3 * pix + 1], p.acc[3 * pix + 2]);
    int s = p.sample_begin;
    bool alive = inside && s < p.sample_end;
    bool hasExt = false, hasShadow = false, fresh = true;
    f3 ro = mk3(0.0f), rd = mk3(1.0f), rinv = mk3(1.0f);       // extension ray (+ 1/d, computed at full wave width)
    f3 so = mk3(0.0f), sd = mk3(1.0f);                          // shadow ray
    float stmax = 0.0f;
    f3 beta = mk3(1.0f), rad = mk3(0.0f), pend = mk3(0.0f);
    Rng rng; rng.state = 0;
    int depth = 0, pendNf = 0;                      // pendNf: components of beta that were inf/NaN when pend was formed
    bool shOccluded = false;
    HitRec hit; hit.t = 0.0f; hit.prim = -1; hit.b1 = hit.b2 = 0.0f;

    while (true) {
        // =================== shade (uses the two results of the previous round) ===================
        if (alive) {
            if (hasShadow) {                                   // integrator.cpp:150-165
                if (!shOccluded) rad = rad + pend;
                else rad = poisonNonFinite(rad, pendNf);     // occluded: integrateMIS adds beta * {} (integrator.cpp:168,195)
                hasShadow = false;
            }
            if (hasExt) {
                hasExt = false;
                if (hit.prim < 0) {                            // integrator.cpp:183-187
                    rad = rad + beta * a3(sc.sky);
                } else if (depth++ != p.max_depth) {           // integrator.cpp:191
                    const Surface sf = makeSurface(sc.shade, hit, ro, rd);
                    const DMaterial &mat = sc.materials[sf.material];
                    const f3 wo = -rd;
                    if (sc.num_lights > 0) {                   // sampleLights integrator.cpp:134-169
                        const uint32_t idx = rng.sampleRange(sc.num_lights - 1);
                        const DLight &light = sc.lights[idx];
                        (void) rng.f(); (void) rng.f();
                        LightSample ls;
                        if (lightSample(light, sf.point, ls)) {
                            so = sf.point + sf.normal * RAY_EPSILON;
                            sd = ls.wi;
                            stmax = len(sf.point - ls.p) - RAY_EPSILON;
                            f3 f; float pb;
                            evalPdfBxdf<MASK>(ctx, mat, sf.normal, sf.uv, wo, ls.wi, f, pb);
                            f = f * absdot(ls.wi, sf.normal);
                            const float pl = 1.0f / (float) sc.num_lights * ls.pdf;
                            const float misWeight = powerHeuristic(1.0f, pl, 1.0f, pb);   // also for delta lights (Q10)
                            pend = beta * (misWeight * f * ls.radiance / pl);
                            pendNf = nonFiniteMask(beta);
                            hasShadow = true;
                        }
                    }
                    const float u = rng.f();
                    f2 u2; u2.x = rng.f(); u2.y = rng.f();
                    BSample bs;
                    if (COUNT) cnt.n_shade++;
                    if (sampleBxdf<MASK>(ctx, mat, sf.normal, sf.uv, wo, u, u2, bs)) {
                        if (bs.pdf > 0.0f) beta = beta * (bs.f * absdot(bs.wi, sf.normal) / bs.pdf);
                        ro = sf.point + bs.wi * RAY_EPSILON;  // integrator.cpp:212
                        rd = bs.wi;
                        hasExt = true;
                    }
                }
            }
            if (!hasExt && !hasShadow) {                       // nothing of the path is in flight: film update, next stratum
                if (!fresh) {
                    f3 c = rad;                                // camera.cpp:110-112
                    if (c.x > 1.0f) c.x = 1.0f;
                    if (c.y > 1.0f) c.y = 1.0f;
                    if (c.z > 1.0f) c.z = 1.0f;
                    acc = acc + c;                             // image.hpp:82-86
                    ++s;
                }
                fresh = false;
                if (s < p.sample_end) {
                    rng.seed(row, col, (uint32_t) s + 1u);     // camera.cpp:101
                    cameraRay(p.cam, col, row, s, rng, ro, rd);
                    beta = mk3(1.0f); rad = mk3(0.0f); depth = 0;
                    hasExt = true;
                    if (COUNT) cnt.n_camera++;
                } else alive = false;
            }
            if (hasExt) rinv = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
        }
        if (__ballot(hasShadow || hasExt) == 0ull) break;      // wave-uniform: every lane is done

        // =================== trace: shadow ray, then extension ray, in one loop ===================
        // stage: 0 = on the shadow ray, 1 = on the extension ray, 2 = nothing (left) to trace
        int stage = hasShadow ? 0 : (hasExt ? 1 : 2);
        f3 o = stage == 0 ? so : ro, d = stage == 0 ? sd : rd;
        f3 inv = stage == 0 ? mk3(1.0f / sd.x, 1.0f / sd.y, 1.0f / sd.z) : rinv;
        float tmin = stage == 0 ? 0.0f : 0.001f, tmax = stage == 0 ? stmax : __builtin_inff();
        int negmask = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
        int cur = stage < 2 ? negmask * sc.num_nodes : -1, leafOff = 0, leafN = 0;      // leafN: count | last-record flag
        bool hitAny = false;
        HitRec rec; rec.t = 0.0f; rec.prim = -1; rec.b1 = rec.b2 = 0.0f;
        bool needSetup = stage < 2;                            // validate (regular / empty scene) the ray just loaded
        while (true) {
            // ---- ray bookkeeping: finish the current ray, load the next one ----
            if (stage < 2 && (needSetup || (cur < 0 && leafN == 0))) {
                if (!needSetup) {
                    // current ray finished: keep its result, move on
                    if (stage == 0) { shOccluded = hitAny; stage = hasExt ? 1 : 2; }
                    else { hit = rec; if (!hitAny) hit.prim = -1; stage = 2; }
                    if (stage == 1) {
                        o = ro; d = rd; inv = rinv; tmin = 0.001f; tmax = __builtin_inff();
                        negmask = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
                        cur = negmask * sc.num_nodes; hitAny = false; rec.prim = -1; rec.t = 0.0f; rec.b1 = rec.b2 = 0.0f;
                        needSetup = true;
                    }
                }
                if (needSetup) {
                    needSetup = false;
                    const bool regular = finiteNonZero(inv.x) && finiteNonZero(inv.y) && finiteNonZero(inv.z) &&
                                         fabsf(o.x) < __builtin_inff() && fabsf(o.y) < __builtin_inff() && fabsf(o.z) < __builtin_inff() &&
                                         tmax == tmax;
                    if (sc.num_nodes == 0) { cur = -1; if (COUNT) { if (stage == 0) cnt.n_any++; else cnt.n_closest++; } }
                    else if (!regular) {
                        // axis-parallel / non-finite rays: exact slab test, traced to the end right here (rare)
                        if (stage == 0) hitAny = traverseThreaded<true, COUNT, false>(src, sc.num_nodes, o, d, inv, negmask, tmin, tmax, rec, cnt);
                        else            hitAny = traverseThreaded<false, COUNT, false>(src, sc.num_nodes, o, d, inv, negmask, tmin, tmax, rec, cnt);
                        cur = -1;
                    } else if (COUNT) { if (stage == 0) cnt.n_any++; else cnt.n_closest++; }
                }
            }
            if (__ballot(stage < 2) == 0ull) break;            // wave-uniform
            if (__ballot(stage < 2 && cur < 0 && leafN == 0) != 0ull) continue;   // an exact-path ray ended at once: book it

            // ---- interior phase ----
            while (true) {
#pragma unroll
                for (int rep = 0; rep < JTX_STEPS_PER_VOTE; ++rep) {
                    if (leafN == 0 && cur >= 0) {
                        const float4 na = src.tnode(cur, 0);
                        const float4 nb = src.tnode(cur, 1);
                        if (COUNT) { if (stage == 0) cnt.n_nodes_any++; else cnt.n_nodes_closest++; }
                        const bool boxHit = slabRegular(na, nb, o, inv, tmin, tmax);
                        const int w = __float_as_int(nb.w), z = __float_as_int(nb.z);
                        if (boxHit && w != 0) { leafN = w; leafOff = z; }
                        else cur = (boxHit || w != 0) ? (w < 0 ? -1 : cur + 1) : z;
                    }
                }
                const unsigned long long walking = __ballot(leafN == 0 && cur >= 0);
                const unsigned long long parked = __ballot(leafN != 0);
                const unsigned long long ended = __ballot(stage < 2 && leafN == 0 && cur < 0);
                if (walking == 0ull || __popcll(parked) >= JTX_LEAF_VOTE || __popcll(ended) >= JTX_SWITCH_VOTE) break;
            }
            // ---- leaf phase ----
            if (leafN != 0) {
                const int n = leafN & 0xffff;
                for (int i = 0; i < n; ++i) {
                    const int prim = leafOff + i;
                    if (COUNT) { if (stage == 0) cnt.n_tri_any++; else cnt.n_tri_closest++; }
                    float b1, b2, root;
                    if (!triTest(src, prim, o, d, tmin, tmax, b1, b2, root)) continue;
                    hitAny = true;
                    if (stage == 0) break;
                    tmax = root;
                    rec.t = root; rec.prim = prim; rec.b1 = b1; rec.b2 = b2;
                    if (COUNT) cnt.n_accept++;
                }
                cur = ((stage == 0 && hitAny) || leafN < 0) ? -1 : cur + 1;
                leafN = 0;
            }
        }
    }

    if (inside) {
        p.acc[3 * pix] = acc.x; p.acc[3 * pix + 1] = acc.y; p.acc[3 * pix + 2] = acc.z;
        if (p.img) {
            const float invN = (float) p.sample_end;            // currSample + 1 of the last pass (camera.cpp:115)
            p.img[3 * pix] = fToByte(acc.x / invN);
            p.img[3 * pix + 1] = fToByte(acc.y / invN);
            p.img[3 * pix + 2] = fToByte(acc.z / invN);
        }
    }
    if (COUNT) {
        const unsigned v[9] = {cnt.n_camera, cnt.n_closest, cnt.n_any, cnt.n_nodes_closest, cnt.n_tri_closest, cnt.n_accept,
                               cnt.n_nodes_any, cnt.n_tri_any, cnt.n_shade};
        for (int i = 0; i < 9; ++i) {
            unsigned long long t = v[i];
            for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
            if (lane == 0 && t) atomicAdd(&p.counters[i], t);
        }
    }
}

} // namespace jtx

using namespace jtx;

hipError_t jtx_launch_render_fused(const RenderParams &p, int num_owned_tiles, bool count, hipStream_t stream) {
    if (num_owned_tiles <= 0) return hipSuccess;
    const dim3 grid((unsigned) num_owned_tiles * 4u), block(FBLOCK);
    const bool lds = p.scene.lds_threaded != 0;
    size_t shmem = 0;
    if (lds) shmem += ((size_t) 2 * 8 * p.scene.num_nodes + (size_t) 3 * p.scene.num_prims) * sizeof(float4);
    const bool lambert = p.scene.material_mask == MAT_DIFFUSE_ONLY;
#define LAUNCH_F(C, L, M) hipLaunchKernelGGL((k_render_fused<C, L, M>), grid, block, shmem, stream, p)
    if (lambert) {
        if (lds) { if (count) LAUNCH_F(true, true, MAT_DIFFUSE_ONLY); else LAUNCH_F(false, true, MAT_DIFFUSE_ONLY); }
        else     { if (count) LAUNCH_F(true, false, MAT_DIFFUSE_ONLY); else LAUNCH_F(false, false, MAT_DIFFUSE_ONLY); }
    } else {
        if (lds) { if (count) LAUNCH_F(true, true, MAT_ALL); else LAUNCH_F(false, true, MAT_ALL); }
        else     { if (count) LAUNCH_F(true, false, MAT_ALL); else LAUNCH_F(false, false, MAT_ALL); }
    }
#undef LAUNCH_F
    return hipGetLastError();
}
