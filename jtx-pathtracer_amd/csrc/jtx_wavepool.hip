// jtx_wavepool.hip -- the wave-pool integrator: a wavefront path tracer whose queues live in LDS.
//
// One wave owns an 8x8 pixel block; every lane OWNS one pixel and keeps that pixel's path state
// (throughput, radiance, RNG, accumulated strata) in registers -- nothing per path ever goes to HBM.
// But lanes do not trace their own rays.  Each round:
//
//   owner phase  : every lane consumes last round's two results (shadow ray of the previous vertex,
//                  extension ray), shades (light sample + BSDF sample of integrateMIS,
//                  integrator.cpp:171-216) and posts up to two new rays -- a shadow ray and the next
//                  extension ray -- into the wave's ray pool in LDS (SoA-in-LDS, 128 rays).
//   worker phase : wave-ballot compaction turns the posted rays into a dense list; the 64 lanes then
//                  pull rays from that list dynamically: a lane whose ray ends early takes the next
//                  one, so the lanes stay busy although ray lengths differ by an order of magnitude
//                  and although shadow and extension rays are mixed.  Hits go back through LDS.
//
// That is the persistent-threads wavefront scheme (SoA ray buffers, ballot compaction, LDS
// traversal stack, LDS-staged BVH) scaled down to a wave, so that its queue traffic stays on-chip:
// the HBM-queued variant (jtx_wavefront.hip) moves ~300 B per bounce and is HBM-bound on this path.
// A lane whose path ends starts its next stratum at once (in-lane regeneration), strata of a pixel
// are accumulated in order, and every ray's arithmetic is the oracle's: output is bit-identical.
#include "jtx_scene_dev.hpp"
#include "jtx_launch.hpp"

namespace jtx {

constexpr int PBLOCK = 256;
constexpr int POOL = 128;                       // ray slots per wave: 2 per lane
constexpr int POOL_INTS = POOL * 8 + POOL;               // rays (2 float4; the result overwrites the first) + list, in ints

#ifndef JTX_POOL_ASSIGN_VOTE
#define JTX_POOL_ASSIGN_VOTE 8                  // idle lanes that trigger an assignment of new rays
#endif
#ifndef JTX_POOL_LEAF_VOTE
#define JTX_POOL_LEAF_VOTE 16                   // lanes parked on a leaf that end the interior phase
#endif

JD unsigned char wpToByte(float v) {                                       // image.hpp:9-16,47-52
    const float g = v > 0.0f ? sqrtf(v) : 0.0f;
    const float c = clampf(g, 0.0f, 0.999f);
    return (unsigned char) (int) (255.999f * c);
}

template <bool COUNT, bool LDS_SCENE>
__global__ void __launch_bounds__(PBLOCK) k_render_wavepool(RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const DevScene &sc = p.scene;
    // LDS: [4 wave pools][8 threaded node orderings][tris]  -- the traversal is stackless
    int *poolAll = smem;
    float4 *lds_tnodes = (float4 *) (poolAll + 4 * POOL_INTS);
    float4 *lds_tris = lds_tnodes + 2 * 8 * sc.num_nodes;
    if (LDS_SCENE) {
        const int nn = 2 * 8 * sc.num_nodes, nt = 3 * sc.num_prims;
        for (int i = threadIdx.x; i < nn; i += PBLOCK) lds_tnodes[i] = sc.tnodes[i];
        for (int i = threadIdx.x; i < nt; i += PBLOCK) lds_tris[i] = sc.tris[i];
        __syncthreads();
    }
    const float4 *tnodes = LDS_SCENE ? lds_tnodes : sc.tnodes;
    const float4 *tris = LDS_SCENE ? lds_tris : sc.tris;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4 *pray = (float4 *) (poolAll + wave * POOL_INTS);       // [POOL][2]
    int *plist = (int *) (pray + 2 * POOL);                       // [POOL]
    const unsigned long long below = (1ull << lane) - 1ull;

    // work mapping: block = 4 waves = 4 consecutive 8x8 sub-blocks of one owned 32x32 tile (camera.cpp:55-64)
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

    // ---- owner state ----
    f3 acc = mk3(0.0f);
    if (inside && p.sample_begin > 0) acc = mk3(p.acc[3 * pix], p.acc[3 * pix + 1], p.acc[3 * pix + 2]);
    int s = p.sample_begin;
    bool alive = inside && s < p.sample_end;
    bool hasExt = false, hasShadow = false, fresh = true;
    f3 ro = mk3(0.0f), rd = mk3(1.0f);              // the extension ray in flight (needed to shade its hit)
    f3 beta = mk3(1.0f), rad = mk3(0.0f), pend = mk3(0.0f);
    Rng rng; rng.state = 0;
    int depth = 0, pendNf = 0;                      // pendNf: components of beta that were inf/NaN when pend was formed

    while (true) {
        // =================== owner phase ===================
        if (alive) {
            if (hasShadow) {                                   // sampleLights' occlusion test came back (integrator.cpp:150-165)
                if (pray[2 * (2 * lane)].w == 0.0f) rad = rad + pend;
                else rad = poisonNonFinite(rad, pendNf);     // occluded: integrateMIS adds beta * {} (integrator.cpp:168,195)
                hasShadow = false;
            }
            if (hasExt) {
                hasExt = false;
                const float4 hv = pray[2 * (2 * lane + 1)];
                const int prim = __float_as_int(hv.w);
                if (prim < 0) {                                // integrator.cpp:183-187
                    rad = rad + beta * a3(sc.sky);
                } else if (depth++ != p.max_depth) {           // integrator.cpp:191
                    HitRec h; h.t = hv.x; h.b1 = hv.y; h.b2 = hv.z; h.prim = prim;
                    const Surface sf = makeSurface(sc.shade, h, ro, rd);
                    const DMaterial &mat = sc.materials[sf.material];
                    const f3 wo = -rd;
                    if (sc.num_lights > 0) {                   // sampleLights integrator.cpp:134-169
                        const uint32_t idx = rng.sampleRange(sc.num_lights - 1);
                        const DLight &light = sc.lights[idx];
                        (void) rng.f(); (void) rng.f();
                        LightSample ls;
                        if (lightSample(light, sf.point, ls)) {
                            const f3 so = sf.point + sf.normal * RAY_EPSILON;
                            const float lDist = len(sf.point - ls.p);
                            f3 f; float pb;
                            evalPdfBxdf(ctx, mat, sf.normal, sf.uv, wo, ls.wi, f, pb);
                            f = f * absdot(ls.wi, sf.normal);
                            const float pl = 1.0f / (float) sc.num_lights * ls.pdf;
                            const float misWeight = powerHeuristic(1.0f, pl, 1.0f, pb);   // also for delta lights (Q10)
                            pend = beta * (misWeight * f * ls.radiance / pl);
                            pendNf = nonFiniteMask(beta);             // added if the shadow ray gets through
                            pray[2 * (2 * lane) + 0] = make_float4(so.x, so.y, so.z, lDist - RAY_EPSILON);
                            pray[2 * (2 * lane) + 1] = make_float4(ls.wi.x, ls.wi.y, ls.wi.z, __int_as_float(1));
                            hasShadow = true;
                        }
                    }
                    const float u = rng.f();
                    f2 u2; u2.x = rng.f(); u2.y = rng.f();
                    BSample bs;
                    if (COUNT) cnt.n_shade++;
                    if (sampleBxdf(ctx, mat, sf.normal, sf.uv, wo, u, u2, bs)) {
                        if (bs.pdf > 0.0f) beta = beta * (bs.f * absdot(bs.wi, sf.normal) / bs.pdf);
                        ro = sf.point + bs.wi * RAY_EPSILON;  // integrator.cpp:212
                        rd = bs.wi;
                        hasExt = true;
                    }
                }
            }
            if (!hasExt && !hasShadow) {
                // the path is complete (nothing of it is in flight any more): film update, next stratum
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
            if (hasExt) {
                pray[2 * (2 * lane + 1) + 0] = make_float4(ro.x, ro.y, ro.z, __builtin_inff());
                pray[2 * (2 * lane + 1) + 1] = make_float4(rd.x, rd.y, rd.z, __int_as_float(0));
            }
        }
        // =================== compaction: dense list of posted rays ===================
        const unsigned long long mS = __ballot(hasShadow), mE = __ballot(hasExt);
        const int nS = __popcll(mS), nRays = nS + __popcll(mE);
        if (nRays == 0) break;                                 // wave-uniform: every lane is done
        if (hasShadow) plist[__popcll(mS & below)] = 2 * lane;
        if (hasExt) plist[nS + __popcll(mE & below)] = 2 * lane + 1;
        __builtin_amdgcn_wave_barrier();

        // =================== worker phase ===================
        UTIL(if (COUNT) cnt.it_calls++;)
        int next = 0;                                          // wave-uniform: first unassigned list entry
        int ray = -1, cur = -1, leafOff = 0, leafN = 0, negmask = 0;      // leafN: count | last-record flag
        bool isAny = false, hitAny = false;
        f3 o = mk3(0.0f), d = mk3(1.0f), inv = mk3(1.0f);
        float tmin = 0.0f, tmax = 0.0f;
        HitRec rec; rec.t = 0.0f; rec.prim = -1; rec.b1 = rec.b2 = 0.0f;
        while (true) {
            // ---- hand unassigned rays to idle lanes ----
            const unsigned long long idle = __ballot(ray < 0);
            if (next < nRays && (__popcll(idle) >= JTX_POOL_ASSIGN_VOTE || idle == ~0ull)) {
                const int k = __popcll(idle & below);
                if (ray < 0 && next + k < nRays) {
                    ray = plist[next + k];
                    const float4 r0 = pray[2 * ray], r1 = pray[2 * ray + 1];
                    o = mk3(r0.x, r0.y, r0.z); d = mk3(r1.x, r1.y, r1.z);
                    tmax = r0.w; isAny = __float_as_int(r1.w) != 0;
                    tmin = isAny ? 0.0f : 0.001f;
                    inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                    negmask = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
                    const bool regular = finiteNonZero(inv.x) && finiteNonZero(inv.y) && finiteNonZero(inv.z) &&
                                         fabsf(o.x) < __builtin_inff() && fabsf(o.y) < __builtin_inff() && fabsf(o.z) < __builtin_inff() &&
                                         tmax == tmax;
                    cur = negmask * sc.num_nodes; leafN = 0; hitAny = false; rec.prim = -1; rec.t = 0.0f; rec.b1 = rec.b2 = 0.0f;
                    if (sc.num_nodes == 0) cur = -1;
                    else if (!regular) {
                        // axis-parallel / non-finite rays: the exact slab test, traced to the end right here (rare)
                        GlobalSrc src; src.tnodes = tnodes; src.tris = tris;
                        if (isAny) hitAny = traverseThreaded<true, COUNT, false>(src, sc.num_nodes, o, d, inv, negmask, tmin, tmax, rec, cnt);
                        else       hitAny = traverseThreaded<false, COUNT, false>(src, sc.num_nodes, o, d, inv, negmask, tmin, tmax, rec, cnt);
                        cur = -1;
                    }
                    if (COUNT && (regular || sc.num_nodes == 0)) { if (isAny) cnt.n_any++; else cnt.n_closest++; }
                }
                const int nIdle = __popcll(idle);
                next += nIdle < nRays - next ? nIdle : nRays - next;
            }
            if (__ballot(ray >= 0) == 0ull) break;             // list exhausted and every lane retired

            UTIL(if (COUNT) cnt.it_leaf++;)
            // ---- interior phase: one node per walking lane per iteration ----
            while (true) {
                UTIL(if (COUNT) cnt.it_interior++;)
#pragma unroll
                for (int rep = 0; rep < JTX_STEPS_PER_VOTE; ++rep) {
                    if (ray >= 0 && leafN == 0 && cur >= 0) {
                        const float4 na = tnodes[2 * cur], nb = tnodes[2 * cur + 1];
                        if (COUNT) { if (isAny) cnt.n_nodes_any++; else cnt.n_nodes_closest++; }
                        const bool boxHit = slabRegular(na, nb, o, inv, tmin, tmax);
                        const int w = __float_as_int(nb.w), z = __float_as_int(nb.z);
                        if (boxHit && w != 0) { leafN = w; leafOff = z; }
                        else cur = (boxHit || w != 0) ? (w < 0 ? -1 : cur + 1) : z;
                    }
                }
                const unsigned long long walking = __ballot(ray >= 0 && leafN == 0 && cur >= 0);
                const unsigned long long parked = __ballot(leafN != 0);
                const unsigned long long free_ = __ballot(leafN == 0 && cur < 0);     // finished or never assigned
                if (walking == 0ull || __popcll(parked) >= JTX_POOL_LEAF_VOTE ||
                    (next < nRays && __popcll(free_) >= JTX_POOL_ASSIGN_VOTE)) break;
            }

            // ---- leaf phase ----
            if (leafN != 0) {
                GlobalSrc src; src.tris = tris;
                const int n = leafN & 0xffff;
                for (int i = 0; i < n; ++i) {
                    const int prim = leafOff + i;
                    if (COUNT) { if (isAny) cnt.n_tri_any++; else cnt.n_tri_closest++; }
                    float b1, b2, root;
                    if (!triTest(src, prim, o, d, tmin, tmax, b1, b2, root)) continue;
                    hitAny = true;
                    if (isAny) break;
                    tmax = root;
                    rec.t = root; rec.prim = prim; rec.b1 = b1; rec.b2 = b2;
                    if (COUNT) cnt.n_accept++;
                }
                cur = ((isAny && hitAny) || leafN < 0) ? -1 : cur + 1;
                leafN = 0;
            }

            // ---- retire ----
            if (ray >= 0 && cur < 0 && leafN == 0) {
                if (isAny) pray[2 * ray] = make_float4(0.0f, 0.0f, 0.0f, hitAny ? 1.0f : 0.0f);
                else pray[2 * ray] = make_float4(rec.t, rec.b1, rec.b2, __int_as_float(hitAny ? rec.prim : -1));
                ray = -1;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (inside) {
        p.acc[3 * pix] = acc.x; p.acc[3 * pix + 1] = acc.y; p.acc[3 * pix + 2] = acc.z;
        if (p.img) {
            const float invN = (float) p.sample_end;            // currSample + 1 of the last pass (camera.cpp:115)
            p.img[3 * pix] = wpToByte(acc.x / invN);
            p.img[3 * pix + 1] = wpToByte(acc.y / invN);
            p.img[3 * pix + 2] = wpToByte(acc.z / invN);
        }
    }
#ifdef JTX_PROFILE_UTIL
    if (COUNT) {
        unsigned a = cnt.it_interior, b = cnt.it_leaf, c = cnt.it_calls;
        for (int off = 32; off > 0; off >>= 1) { a = max(a, __shfl_down(a, off, 64)); b = max(b, __shfl_down(b, off, 64)); c = max(c, __shfl_down(c, off, 64)); }
        if (lane == 0) { atomicAdd(&p.counters[20], (unsigned long long) a); atomicAdd(&p.counters[21], (unsigned long long) b); atomicAdd(&p.counters[22], (unsigned long long) c); }
    }
#endif
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

hipError_t jtx_launch_render_wavepool(const RenderParams &p, int num_owned_tiles, bool count, hipStream_t stream) {
    if (num_owned_tiles <= 0) return hipSuccess;
    const dim3 grid((unsigned) num_owned_tiles * 4u), block(PBLOCK);
    const bool lds = p.scene.lds_threaded != 0;
    size_t shmem = (size_t) 4 * POOL_INTS * sizeof(int);
    if (lds) shmem += ((size_t) 2 * 8 * p.scene.num_nodes + (size_t) 3 * p.scene.num_prims) * sizeof(float4);
    if (lds) {
        if (count) hipLaunchKernelGGL((k_render_wavepool<true, true>), grid, block, shmem, stream, p);
        else       hipLaunchKernelGGL((k_render_wavepool<false, true>), grid, block, shmem, stream, p);
    } else {
        if (count) hipLaunchKernelGGL((k_render_wavepool<true, false>), grid, block, shmem, stream, p);
        else       hipLaunchKernelGGL((k_render_wavepool<false, false>), grid, block, shmem, stream, p);
    }
    return hipGetLastError();
}
