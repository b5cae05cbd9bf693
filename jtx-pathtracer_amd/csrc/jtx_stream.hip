// jtx_stream.hip -- k_render_stream: the RE-ENTRANT path kernel (the timed kernel of integrator 1 since round 2).
//
// Why.  In k_render_paths a wave traces one ray per lane to the end before anybody shades: counters of that kernel
// (tools/tools_wide_stats.py, tools/tools_util.py) show 45-55 % of all lane-iterations of the traversal loops spent
// by lanes whose own ray is FINISHED, waiting for the wave's longest one (22 % of the node iterations run with one or
// two walking lanes).  Here no lane ever waits for another lane's ray.  A lane is in one of three stages
//      IDLE   no path                      -> the hand-out gives it the next path of the wave's chunk
//      TRACE  its current ray is under way -> node steps / leaf steps of the shared traversal loop
//      READY  its extension ray came back  -> the shade block (integrateMIS's loop body, integrator.cpp:183-213)
// and each trip of the wave's main loop runs ONE block, chosen by ballots: the shade block (+ hand-out) as soon as
// enough lanes are READY / IDLE, the leaf block when enough lanes are parked on a leaf, the node block otherwise.
// Lanes that come out of the shade block walk the same traversal loop as everybody else, from the root.
//
// Two rays per shading event.  The shade block of path vertex k produces BOTH rays that leave the vertex: the
// shadow ray of sampleLights (integrator.cpp:134-169) and the extension ray of sampleBxdf (integrator.cpp:198-212;
// its direction does not depend on the shadow ray's answer, and the shadow ray consumes no random numbers), plus
// C = beta * misWeight * f * L / pl, what the light sample adds if it is unoccluded.  The lane traces the shadow ray,
// adds C (or the reference's beta * {} when occluded) the moment it resolves, then traces the extension ray and
// becomes READY: radiance receives its terms in the reference's order (C_k before anything of vertex k + 1), so the
// per-path radiance -- and with k_resolve_samples' in-order sums the film -- equals integrateMIS's bit for bit.
//
// Traversal blocks are those of jtx_scene_dev.hpp (threaded binary records from LDS / HBM, 8-ary quantised nodes),
// with the closest / any distinction as a per-lane flag.  Rays the fast slab forms cannot take (axis-parallel,
// non-finite, out of the wide nodes' range) are traced to the end on the exact binary records when they start.
// Compiled with -ffp-contract=off; uncounted launches only (the counting launches stay on k_render_pixels).
#include "jtx_scene_dev.hpp"
#include "jtx_launch.hpp"

namespace jtx {

#ifndef JTX_ST_OCC_LDS
#define JTX_ST_OCC_LDS 6          // waves per SIMD: LDS-resident scenes
#endif
#ifndef JTX_ST_OCC_WIDE
#define JTX_ST_OCC_WIDE 5         // ... 8-ary nodes in HBM
#endif
#ifndef JTX_ST_OCC_WIDE_ALL
#define JTX_ST_OCC_WIDE_ALL 4     // ... with every BxDF in the shade block
#endif
#ifndef JTX_ST_SHADE_VOTE
#define JTX_ST_SHADE_VOTE 20      // READY + IDLE lanes that call the shade block
#endif
#ifndef JTX_ST_BEGIN_VOTE
#define JTX_ST_BEGIN_VOTE 12      // lanes with a new ray that call the begin block
#endif
#ifndef JTX_ST_LEAF_VOTE
#define JTX_ST_LEAF_VOTE 16       // parked lanes that call the leaf block
#endif
#ifndef JTX_ST_STEPS
#define JTX_ST_STEPS 4            // binary node steps per vote
#endif
#ifndef JTX_ST_WSTEPS
#define JTX_ST_WSTEPS 1           // wide node steps per vote
#endif

enum { ST_IDLE = 0, ST_TRACE = 1, ST_READY = 2, ST_FRESH = 3 };   // FRESH: holds a new ray (o, d, tmax, flags) that has not started yet
enum { SF_SHADOW = 1,            // the current ray is the shadow ray of sampleLights (anyHit); else closestHit
       SF_NEXT = 2,              // an extension ray waits behind the shadow ray (else the path ends with it)
       SF_NF_SHIFT = 4 };        // bits 4-6: components of beta that were inf / NaN at the vertex (beta * {} is NaN there)

constexpr int ST_BLOCK = JTX_RP_BLOCK;

// ---- node step of the 8-ary traversal with the closest / any choice per lane (wideNodeStep<ORDERED>, jtx_scene_dev.hpp) ----
JD void wideNodeStepRT(const uint4 *__restrict__ wide, uint2 *stk, int stride, const WideRay &r, WideState &ws, bool ordered) {
    if ((ws.gbits & 0xffu) == 0u) {                          // group exhausted: pop
        if (ws.sp == 0) { ws.done = true; return; }
        --ws.sp; const uint2 e = stk[ws.sp * stride]; ws.gbase = e.x; ws.gbits = e.y;
    }
    const unsigned live = ws.gbits & 0xffu;
    const int k = ordered ? __builtin_ctz(live) : 31 - __builtin_clz(live);   // anyHit: slot order, leaves first
    ws.gbits &= ~(1u << k);
    const unsigned slot = (ws.gbits >> (8 + 3 * k)) & 7u;
    const unsigned ni = ws.gbase >> 28, base = ws.gbase & 0x0fffffffu;
    if (slot >= ni) { ws.pendLeaf = (int) (base + 6u * ni + 2u * (slot - ni)); return; }
    const unsigned a = base + 6u * slot;
    if (ws.gbits & 0xffu) { stk[ws.sp * stride] = make_uint2(ws.gbase, ws.gbits); ++ws.sp; }
    const uint4 n0 = wide[a], n2 = wide[a + 1], n3 = wide[a + 2], n4 = wide[a + 3];
    const unsigned *tail = (const unsigned *) (wide + a + 4);
    const int pbit = 24 * r.negmask;
    const unsigned cbase = tail[0];
    const unsigned plo = tail[1 + (pbit >> 5)], phi = tail[2 + (pbit >> 5)];
    const f3 o = r.o, inv = r.inv;
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
    unsigned hits = 0u;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int w = s >> 2, b = s & 3;
        const float t0 = fmaxf(fmaxf(__fmaf_rn(ubyteToFloat(nxq[w], b), axx, bnx), __fmaf_rn(ubyteToFloat(nyq[w], b), ayy, bny)),
                               fmaxf(__fmaf_rn(ubyteToFloat(nzq[w], b), azz, bnz), r.tmin));
        const float t1 = fminf(fminf(__fmaf_rn(ubyteToFloat(fxq[w], b), axx, bfx), __fmaf_rn(ubyteToFloat(fyq[w], b), ayy, bfy)),
                               fminf(__fmaf_rn(ubyteToFloat(fzq[w], b), azz, bfz), r.tmax));
        hits |= (t0 <= t1 ? 1u : 0u) << s;
    }
    const unsigned perm = ordered ? (__funnelshift_r(plo, phi, pbit & 31) & 0x00ffffffu) : 0x00fac688u;
    const unsigned nchild = n0.w >> 28;
    unsigned pend = 0u;
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) pend |= ((hits >> ((perm >> (3 * k2)) & 7u)) & 1u) << k2;
    pend &= (1u << nchild) - 1u;
    ws.gbase = cbase | (((n0.w >> 24) & 0xfu) << 28);
    ws.gbits = pend | (perm << 8);
}

// ---- exact traversal of one ray on the binary threaded records (rays the fast forms cannot take; rare) ----
template <class Src>
JD bool traverseExactRT(const Src &src, int num_nodes, f3 o, f3 d, f3 inv, int negmask, float tmin, float tmax, bool any, HitRec &rec) {
    int cur = negmask * num_nodes;
    bool hitAnything = false;
    while (cur >= 0) {
        const float4 na = src.tnode(cur, 0), nb = src.tnode(cur, 1);
        const bool boxHit = slabExact(na, nb, o, inv, tmin, tmax);
        const int w = __float_as_int(nb.w), z = __float_as_int(nb.z);
        if (boxHit && w != 0) {
            const int n = w & 0xffff;
            for (int i = 0; i < n; ++i) {
                float b1, b2, root;
                if (!triTest(src, z + i, o, d, tmin, tmax, b1, b2, root)) continue;
                hitAnything = true;
                if (any) break;
                tmax = root; rec.t = root; rec.prim = z + i; rec.b1 = b1; rec.b2 = b2;
            }
            cur = ((any && hitAnything) || w < 0) ? -1 : cur + 1;
        } else cur = (boxHit || w != 0) ? (w < 0 ? -1 : cur + 1) : z;
    }
    return hitAnything;
}

template <int SRC> struct SrcPick { typedef GlobalSrc type; };
template <> struct SrcPick<SRC_LDS> { typedef LdsSrc type; };
template <int SRC> using SrcOf = typename SrcPick<SRC>::type;

template <int SRC, int MASK, int BS>
__global__ void __launch_bounds__(BS, SRC == SRC_WIDE ? (MASK == MAT_DIFFUSE_ONLY ? JTX_ST_OCC_WIDE : JTX_ST_OCC_WIDE_ALL) : JTX_ST_OCC_LDS)
k_render_stream(RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    constexpr bool WIDE = SRC == SRC_WIDE;
    const DevScene &sc = p.scene;
    float4 *lds_tnodes = (float4 *) smem;
    float4 *lds_tris = lds_tnodes + 2 * 8 * sc.num_nodes;
    if (SRC == SRC_LDS) {
        const int nn = 2 * 8 * sc.num_nodes, nt = 3 * sc.num_prims;
        for (int i = threadIdx.x; i < nn; i += BS) lds_tnodes[(i & 1) * (nn >> 1) + (i >> 1)] = sc.tnodes[i];    // LdsSrc: halves apart
        for (int i = threadIdx.x; i < nt; i += BS) lds_tris[i] = sc.tris[i];
        __syncthreads();
    }
    // the binary records (LDS copy or HBM) -- node / triangle source of the binary walk, triangle source of the wide walk
    SrcOf<SRC> src; src.tnodes = sc.tnodes; src.tris = sc.tris;
    if constexpr (SRC == SRC_LDS) { src.tnodes = lds_tnodes; src.tris = lds_tris; src.half = 8 * sc.num_nodes; }
    uint2 *stk = (uint2 *) smem + threadIdx.x;                 // WIDE: this lane's stack column (stride BS)
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int nchunks = p.num_subblocks * p.num_groups;

    // wave-uniform chunk state (k_render_paths' hand-out)
    int next = 0, nunits = 0, row0 = 0, col0 = 0, slot0 = 0, sBegin = 0;
    bool exhausted = false;

    // ---- per-lane state ----
    int stage = ST_IDLE, flags = 0;
    f3 o = mk3(0.0f), d = mk3(1.0f), inv = mk3(1.0f);          // current ray
    float tmax = 0.0f; int negmask = 0;
    HitRec rec; rec.t = 0.0f; rec.prim = -1; rec.b1 = rec.b2 = 0.0f;
    bool hitAny = false;
    int cur = -1, leafW = 0, leafOff = 0;                      // binary walk
    WideState ws; ws.start(); ws.done = true;                  // wide walk
    f3 no = mk3(0.0f), nd = mk3(1.0f);                         // extension ray waiting behind the shadow ray
    f3 beta = mk3(1.0f), radiance = mk3(0.0f), pendC = mk3(0.0f);
    Rng rng; rng.state = 0u;
    int depth = 0, s = 0, slot = 0;

#ifdef JTX_PROFILE_STREAM
    unsigned long long st_trips[4] = {0, 0, 0, 0}, st_lanes[4] = {0, 0, 0, 0};   // node / leaf / shade / begin: trips, active lanes
#define STSTAT(i, mask) { st_trips[i]++; st_lanes[i] += __popcll(mask); }
#else
#define STSTAT(i, mask)
#endif

    while (true) {
        // ================= RETIRE: rays that came back =================
        {
            bool finished;
            if (WIDE) finished = stage == ST_TRACE && ws.done && ws.pendLeaf < 0;
            else      finished = stage == ST_TRACE && leafW == 0 && cur < 0;
            if (finished) {
                if (flags & SF_SHADOW) {
                    // integrator.cpp:194-196: radiance += beta * sampleLights(); an occluded sample is beta * {} (NaN where beta overflowed)
                    if (!hitAny) radiance = radiance + pendC;
                    else radiance = poisonNonFinite(radiance, (flags >> SF_NF_SHIFT) & 7);
                    if (flags & SF_NEXT) {
                        o = no; d = nd; tmax = __builtin_inff(); flags = 0;
                        stage = ST_FRESH;
                    } else {
                        f3 c = radiance;                                      // camera.cpp:110-112
                        if (c.x > 1.0f) c.x = 1.0f;
                        if (c.y > 1.0f) c.y = 1.0f;
                        if (c.z > 1.0f) c.z = 1.0f;
                        p.rad[(size_t) (s - p.sample_begin) * p.rad_stride + slot] = make_float4(c.x, c.y, c.z, 0.0f);
                        stage = ST_IDLE;
                    }
                } else stage = ST_READY;
            }
        }
        bool walking, parked;
        if (WIDE) { walking = stage == ST_TRACE && ws.walking(); parked = stage == ST_TRACE && ws.pendLeaf >= 0; }
        else      { walking = stage == ST_TRACE && leafW == 0 && cur >= 0; parked = stage == ST_TRACE && leafW != 0; }
        const unsigned long long mWalk = __ballot(walking), mPark = __ballot(parked);
        const int nFeed = __popcll(__ballot(stage == ST_READY || (stage == ST_IDLE && !exhausted)));
        const unsigned long long mFresh = __ballot(stage == ST_FRESH);
        const bool nothingUnderWay = (mWalk | mPark) == 0ull;

        if (nFeed < JTX_ST_SHADE_VOTE && mFresh != 0ull && (__popcll(mFresh) >= JTX_ST_BEGIN_VOTE || nothingUnderWay)) {
            // ================= BEGIN BLOCK: new rays start at the root; the ones the fast slab forms cannot take are traced here =================
            STSTAT(3, mFresh)
            bool slow = false;
            if (stage == ST_FRESH) {
                stage = ST_TRACE;
                inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                negmask = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
                hitAny = false; rec.prim = -1;
                cur = negmask * sc.num_nodes; leafW = 0;
                if (WIDE) ws.start();
                if (sc.num_nodes == 0) { cur = -1; if (WIDE) ws.done = true; }   // no geometry: comes back at once
                else {
                    const float tmin = (flags & SF_SHADOW) ? 0.0f : 0.001f;
                    if (WIDE) slow = !wideRayOk(o, inv, tmin, tmax); else slow = !regularRay(o, inv, tmin, tmax);
                }
            }
            if (__builtin_expect(__ballot(slow) != 0ull, 0)) {
                if (slow) {
                    const bool any = (flags & SF_SHADOW) != 0;
                    HitRec h = rec;
                    hitAny = traverseExactRT(src, sc.num_nodes, o, d, inv, negmask, any ? 0.0f : 0.001f, tmax, any, h);
                    if (hitAny && !any) { rec = h; tmax = h.t; }
                    cur = -1; leafW = 0;
                    if (WIDE) { ws.done = true; ws.pendLeaf = -1; }
                }
            }
            continue;                                                         // (rays that came back at once retire at the top)
        }
        if (nFeed >= JTX_ST_SHADE_VOTE || nothingUnderWay) {
            if (nFeed == 0) break;                                            // nothing under way, nothing to start
            STSTAT(2, __ballot(stage == ST_READY))
            // ================= SHADE BLOCK: integrateMIS's loop body for the READY lanes =================
            if (stage == ST_READY) {
                bool ended = false;
                if (!hitAny) {                                                // integrator.cpp:183-187
                    radiance = radiance + beta * a3(sc.sky);
                    ended = true;
                } else if (depth++ == p.max_depth) {                          // integrator.cpp:191
                    ended = true;
                } else {
                    HitRec h = rec; h.t = tmax;
                    const Surface sf = makeSurface(sc.shade, h, o, d);
                    const DMaterial &mat = sc.materials[sf.material];
                    ShadeCtx ctx; ctx.materials = sc.materials; ctx.textures = sc.textures; ctx.texels = sc.texels;
                    const f3 wo = -d;
                    flags = 0;
                    bool haveShadow = false;
                    f3 so = mk3(0.0f), sd = mk3(1.0f); float stmax = 0.0f;
                    if (sc.num_lights > 0) {                                  // sampleLights integrator.cpp:134-169
                        const uint32_t idx = rng.sampleRange(sc.num_lights - 1);
                        const DLight &light = sc.lights[idx];
                        (void) rng.f(); (void) rng.f();
                        LightSample ls;
                        if (lightSample(light, sf.point, ls)) {
                            so = sf.point + sf.normal * RAY_EPSILON;
                            const float lDist = len(sf.point - ls.p);
                            sd = ls.wi; stmax = lDist - RAY_EPSILON;
                            f3 f; float pb;
                            evalPdfBxdf<MASK>(ctx, mat, sf.normal, sf.uv, wo, ls.wi, f, pb);
                            f = f * absdot(ls.wi, sf.normal);
                            const float pl = 1.0f / (float) sc.num_lights * ls.pdf;
                            const float misWeight = powerHeuristic(1.0f, pl, 1.0f, pb);   // applied to delta lights too (Q10)
                            pendC = beta * (misWeight * f * ls.radiance / pl);
                            flags = SF_SHADOW | (nonFiniteMask(beta) << SF_NF_SHIFT);
                            haveShadow = true;
                        }
                    }
                    const float u = rng.f();
                    f2 u2; u2.x = rng.f(); u2.y = rng.f();
                    BSample bs;
                    const bool cont = sampleBxdf<MASK>(ctx, mat, sf.normal, sf.uv, wo, u, u2, bs);
                    if (cont) {
                        if (bs.pdf > 0.0f) beta = beta * (bs.f * absdot(bs.wi, sf.normal) / bs.pdf);
                        no = sf.point + bs.wi * RAY_EPSILON;                  // integrator.cpp:212
                        nd = bs.wi;
                    }
                    if (haveShadow) { o = so; d = sd; tmax = stmax; if (cont) flags |= SF_NEXT; stage = ST_FRESH; }
                    else if (cont)  { o = no; d = nd; tmax = __builtin_inff(); stage = ST_FRESH; }
                    else ended = true;
                }
                if (ended) {
                    f3 c = radiance;                                          // camera.cpp:110-112
                    if (c.x > 1.0f) c.x = 1.0f;
                    if (c.y > 1.0f) c.y = 1.0f;
                    if (c.z > 1.0f) c.z = 1.0f;
                    p.rad[(size_t) (s - p.sample_begin) * p.rad_stride + slot] = make_float4(c.x, c.y, c.z, 0.0f);
                    stage = ST_IDLE;
                }
            }
            // ================= HAND-OUT: the next paths of the wave's chunk to the IDLE lanes =================
            bool need = stage == ST_IDLE && !exhausted;
            while (true) {
                const unsigned long long mask = __ballot(need);
                if (mask == 0ull) break;
                if (next >= nunits) {                                         // chunk used up: fetch the next one
                    int c = 0;
                    if (lane == 0) {
                    c = (int) atomicAdd(p.work, 1u);
                    // cancellation poll: the flag lives in HOST memory (one PCIe read), so only every 64th fetch looks, and the
                    // wave that sees it pushes the chunk counter past the end: every other wave stops at its next fetch
                    if (p.stop && (c & 63) == 0 && c < nchunks && __hip_atomic_load(p.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
                        atomicMax(p.work, 0x40000000u); c = nchunks;
                    }
                }
                    c = __shfl(c, 0, 64);
                    if (c >= nchunks) { exhausted = true; need = false; break; }
                    const int grp = c / p.num_subblocks, sb8 = c - grp * p.num_subblocks;
                    const int owned = sb8 >> 4, sub = sb8 & 15;
                    const int tile = p.tile_rank + owned * p.tile_world;
                    const int trow = tile / p.tiles_x, tcol = tile - trow * p.tiles_x;
                    row0 = trow * 32 + (sub >> 2) * 8; col0 = tcol * 32 + (sub & 3) * 8;
                    slot0 = owned * 1024 + sub * 64;
                    sBegin = p.sample_begin + grp * p.strata_per_group;
                    const int sEnd = sBegin + p.strata_per_group < p.sample_end ? sBegin + p.strata_per_group : p.sample_end;
                    nunits = (sEnd > sBegin && row0 < p.height && col0 < p.width) ? (sEnd - sBegin) * 64 : 0;
                    next = 0;
                    continue;
                }
                const int u = next + __popcll(mask & below);
                next += __popcll(mask);
                if (need && u < nunits) {
                    const int pl = u & 63;
                    const int row = row0 + (pl >> 3), col = col0 + (pl & 7);
                    if (row < p.height && col < p.width) {
                        s = sBegin + (u >> 6); slot = slot0 + pl;
                        rng.seed(row, col, (uint32_t) s + 1u);                // camera.cpp:101
                        cameraRay(p.cam, col, row, s, rng, o, d);
                        beta = mk3(1.0f); radiance = mk3(0.0f); depth = 0; flags = 0;
                        tmax = __builtin_inff();
                        stage = ST_FRESH; need = false;
                    }
                }
            }
        } else if (__popcll(mPark) >= JTX_ST_LEAF_VOTE || mWalk == 0ull) {
            // ================= LEAF BLOCK =================
            STSTAT(1, mPark)
            if (WIDE) {
                if (parked) {
                    WideRay r; r.o = o; r.d = d; r.inv = inv; r.tmin = (flags & SF_SHADOW) ? 0.0f : 0.001f; r.tmax = tmax; r.negmask = negmask;
                    ws.hitAnything = hitAny;
                    wideLeafStep(sc.wide, src, (flags & SF_SHADOW) != 0, r, ws, rec);
                    tmax = r.tmax; hitAny = ws.hitAnything;
                }
            } else if (parked) {
                const bool any = (flags & SF_SHADOW) != 0;
                const float tmin = any ? 0.0f : 0.001f;
                const int n = leafW & 0xffff;
                for (int i = 0; i < n; ++i) {
                    const int prim = leafOff + i;
                    float b1, b2, root;
                    if (!triTest(src, prim, o, d, tmin, tmax, b1, b2, root)) continue;
                    hitAny = true;
                    if (any) break;
                    tmax = root; rec.prim = prim; rec.b1 = b1; rec.b2 = b2;
                }
                cur = ((any && hitAny) || leafW < 0) ? -1 : cur + 1;
                leafW = 0;
            }
        } else {
            // ================= NODE BLOCK =================
            STSTAT(0, mWalk)
            if (WIDE) {
#pragma unroll
                for (int rep = 0; rep < JTX_ST_WSTEPS; ++rep) {
                    if (stage == ST_TRACE && ws.walking()) {
                        WideRay r; r.o = o; r.d = d; r.inv = inv; r.tmin = (flags & SF_SHADOW) ? 0.0f : 0.001f; r.tmax = tmax; r.negmask = negmask;
                        wideNodeStepRT(sc.wide, stk, BS, r, ws, !(flags & SF_SHADOW));
                    }
                }
            } else {
                const float tmin = (flags & SF_SHADOW) ? 0.0f : 0.001f;
#pragma unroll
                for (int rep = 0; rep < JTX_ST_STEPS; ++rep) {
                    if (stage == ST_TRACE && leafW == 0 && cur >= 0) {
                        const float4 na = src.tnode(cur, 0), nb = src.tnode(cur, 1);
                        const bool boxHit = slabRegular(na, nb, o, inv, tmin, tmax);
                        const int w = __float_as_int(nb.w), z = __float_as_int(nb.z);
                        if (boxHit && w != 0) { leafW = w; leafOff = z; }
                        else cur = (boxHit || w != 0) ? (w < 0 ? -1 : cur + 1) : z;
                    }
                }
            }
        }

    }
#ifdef JTX_PROFILE_STREAM
    if (p.counters && lane == 0) for (int i = 0; i < 4; ++i) { atomicAdd(&p.counters[32 + i], st_trips[i]); atomicAdd(&p.counters[36 + i], st_lanes[i]); }
#endif
}

} // namespace jtx

using namespace jtx;

hipError_t jtx_launch_render_stream(const RenderParams &p, int num_owned_tiles, int num_cus, hipStream_t stream) {
    if (num_owned_tiles <= 0) return hipSuccess;
    const bool lds = p.scene.lds_threaded != 0;
    const bool wide = !lds && p.scene.wide != nullptr;
    constexpr int SMALL = 64;
    const int bs = lds ? ST_BLOCK : SMALL;
    const bool lambert = p.scene.material_mask == MAT_DIFFUSE_ONLY;
    size_t shmem = 0;
    if (wide) shmem = (size_t) p.scene.wide_depth * bs * sizeof(uint2);
    else if (lds) shmem = ((size_t) 2 * 8 * p.scene.num_nodes + (size_t) 3 * p.scene.num_prims) * sizeof(float4);
    const int occ = wide ? (lambert ? JTX_ST_OCC_WIDE : JTX_ST_OCC_WIDE_ALL) : JTX_ST_OCC_LDS;
    long waves = (long) num_cus * 4 * occ;
    const long chunks = (long) p.num_subblocks * p.num_groups;
    if (waves > chunks) waves = chunks;
    const dim3 grid((unsigned) ((waves * 64 + bs - 1) / bs)), block(bs);
#define LAUNCH_ST(L, M, B) hipLaunchKernelGGL((k_render_stream<L, M, B>), grid, block, shmem, stream, p)
    if (lambert) { if (lds) LAUNCH_ST(SRC_LDS, MAT_DIFFUSE_ONLY, ST_BLOCK); else if (wide) LAUNCH_ST(SRC_WIDE, MAT_DIFFUSE_ONLY, SMALL); else LAUNCH_ST(SRC_GLOBAL, MAT_DIFFUSE_ONLY, SMALL); }
    else         { if (lds) LAUNCH_ST(SRC_LDS, MAT_ALL, ST_BLOCK); else if (wide) LAUNCH_ST(SRC_WIDE, MAT_ALL, SMALL); else LAUNCH_ST(SRC_GLOBAL, MAT_ALL, SMALL); }
#undef LAUNCH_ST
    return hipGetLastError();
}
