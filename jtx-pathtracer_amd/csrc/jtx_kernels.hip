// jtx_kernels.hip -- gfx950 kernels of the path-tracing core and their launchers.
//
//  k_render_pixels   : pixel-persistent integrator.  One lane owns one pixel and runs all of its
//                      strata in the reference's sample order, so the float accumulation order of
//                      AccumulationBuffer::updatePixel (image.hpp:82-86) is preserved with no
//                      atomics and no resolve pass; a lane whose path ends starts its next stratum
//                      at once (in-lane path regeneration), which keeps all 64 lanes of a wave busy
//                      although path lengths differ.  One wave = one 8x8 pixel block.
//                      Used by the counting launches (and JTX_DYNAMIC_PATHS=0).
//  k_render_paths    : the timed kernel.  Persistent waves fetch (8x8 pixel block, strata group) chunks and hand
//                      their paths to whichever lane is free; per-path radiance -> rad[stratum][pixel].
//  k_resolve_samples : adds the per-path radiances to the film in sample order (same float sums).
//  k_resolve_progressive : the same for a PROGRESSIVE launch (k_render_paths<.., PROG>: all passes of a frame in one launch), pass after pass,
//                      beside the path kernel.
//  k_*_batch         : per-ray / per-sample entry points used by the parity tests.
//
// Compiled with -ffp-contract=off: results must equal the CPU oracle bit for bit.
#include "jtx_scene_dev.hpp"
#include "jtx_launch.hpp"
#include <cstdlib>

namespace jtx {

// node steps per scheduling vote of the 8-ary walk in the path kernel: Lambert-only instance closest / any, all-BxDF instance closest / any
// (final build of round 5, ms per frame: C3 with 1,1 / 2,1 / 1,2 / 2,2 / 3,3 = 283.4 / 276.9 / 278.4 / 275.1 / 282.7; C5 with 1,1 / 2,1 / 1,2 = 263.1 / 266.5 / 265.1)
#ifndef JTX_WIDE_STEPS_LC
#define JTX_WIDE_STEPS_LC 2
#endif
#ifndef JTX_WIDE_STEPS_LA
#define JTX_WIDE_STEPS_LA 2
#endif
#ifndef JTX_WIDE_STEPS_MC
#define JTX_WIDE_STEPS_MC 1
#endif
#ifndef JTX_WIDE_STEPS_MA
#define JTX_WIDE_STEPS_MA 1
#endif
#ifndef JTX_RP_BLOCK
#define JTX_RP_BLOCK 256
#endif
#ifndef JTX_RP_OCC
#define JTX_RP_OCC 7
#endif
#ifndef JTX_WIDE_OCC
#define JTX_WIDE_OCC 8          // waves per SIMD of the wide-traversal instances
#endif
#ifndef JTX_NUM_SGPR
#define JTX_NUM_SGPR 0          // > 0: amdgpu_num_sgpr on k_render_paths (the compiler budgets 80 SGPRs by itself and spills 60-80 into VGPR lanes)
#endif
#if JTX_NUM_SGPR > 0
#define JTX_SGPR_ATTR __attribute__((amdgpu_num_sgpr(JTX_NUM_SGPR)))
#else
#define JTX_SGPR_ATTR
#endif
constexpr int BLOCK = JTX_RP_BLOCK;          // threads per workgroup of the render / batch kernels
constexpr int WAVES_PER_BLOCK = BLOCK / 64;
constexpr int BLOCKS_PER_TILE = 16 / WAVES_PER_BLOCK;   // a 32x32 tile = 16 wave-sized 8x8 pixel blocks

// ---- LDS carve of the stackless kernels: [8 threaded node orderings][tris] (16-B aligned), no stack ----
JD void stageScene(const DevScene &sc, float4 *lds_tnodes, float4 *lds_tris) {
    const int nn = 2 * 8 * sc.num_nodes, nt = 3 * sc.num_prims;
    for (int i = threadIdx.x; i < nn; i += BLOCK) lds_tnodes[(i & 1) * (nn >> 1) + (i >> 1)] = sc.tnodes[i];   // LdsSrc: halves apart
#if JTX_LDS_PLANES
    for (int i = threadIdx.x; i < nt; i += BLOCK) lds_tris[(i % 3) * sc.num_prims + i / 3] = sc.tris[i];   // LdsSrc / LeafSrc: three planes
#else
    for (int i = threadIdx.x; i < nt; i += BLOCK) lds_tris[i] = sc.tris[i];
#endif
    __syncthreads();
}

JD void waveAddCounters(unsigned long long *g, const Counters9 &c) {
    const unsigned v[9] = {c.n_camera, c.n_closest, c.n_any, c.n_nodes_closest, c.n_tri_closest, c.n_accept,
                           c.n_nodes_any, c.n_tri_any, c.n_shade};
    for (int i = 0; i < 9; ++i) {
        unsigned long long s = v[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&g[i], s);
    }
    const unsigned w[14] = {c.n_shade_t[0], c.n_shade_t[1], c.n_shade_t[2], c.n_shade_t[3], c.n_shade_t[4], c.n_shade_t[5], c.n_shade_t[6],
                            c.n_eval_t[0], c.n_eval_t[1], c.n_eval_t[2], c.n_eval_t[3], c.n_eval_t[4], c.n_eval_t[5], c.n_eval_t[6]};
    for (int i = 0; i < 14; ++i) {
        unsigned long long s = w[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&g[i < 7 ? CNT_SHADE_T + i : CNT_EVAL_T + i - 7], s);
    }
}

// One bounce of integrateMIS (integrator.cpp:171-216) for the lane's current path.  Returns true
// when the path is finished (radiance final).
struct PathState {
    f3 o, d, beta, radiance;
    Rng rng;
    int depth;
    JTX_PROF_PATH_FIELDS         // diagnostic builds only (jtx_profile.hpp)
};

// -> false: the path goes on with its next extension ray; true: finished (radiance final)
constexpr bool BOUNCE_NEXT = false, BOUNCE_DONE = true;
template <bool COUNT, int MASK, class Src>
JD bool pathBounce(const DevScene &sc, const Src &src, int maxDepth, PathState &ps, Counters9 &cnt) {
    HitRec h;
    PH_DECL
    const bool hit = traverseNoStack<false, COUNT>(src, sc.num_nodes, ps.o, ps.d, 0.001f, __builtin_inff(), h, cnt);
    PH(0)
    if (!hit) {                                                       // integrator.cpp:183-187
        ps.radiance = ps.radiance + ps.beta * a3(sc.sky);
        return BOUNCE_DONE;
    }
    if (ps.depth++ == maxDepth) return BOUNCE_DONE;                           // integrator.cpp:191
    const Surface sf = makeSurface(sc.shade, h, ps.o, ps.d);
    const DMaterial &mat = sc.materials[sf.material];
    ShadeCtx ctx; ctx.materials = sc.materials; ctx.textures = sc.textures; ctx.texels = sc.texels;
    const f3 wo = -ps.d;
    if (sc.num_lights > 0) {                                          // sampleLights integrator.cpp:134-169
        const uint32_t idx = ps.rng.sampleRange(sc.num_lights - 1);
        const DLight &light = sc.lights[idx];
        (void) ps.rng.f(); (void) ps.rng.f();
        LightSample ls;
        if (lightSample(light, sf.point, ls)) {
            const f3 sOrigin = sf.point + sf.normal * RAY_EPSILON;
            const float lDist = len(sf.point - ls.p);
            HitRec dummy;
            PH(1)
            const bool occluded = traverseNoStack<true, COUNT>(src, sc.num_nodes, sOrigin, ls.wi, 0.0f, lDist - RAY_EPSILON, dummy, cnt);
            PH(2)
            if (!occluded) {
                f3 f; float pb;
                if (COUNT) countClass(cnt.n_eval_t, bxdfClass(mat));
                evalPdfBxdf<MASK>(ctx, mat, sf.normal, sf.uv, wo, ls.wi, f, pb);
                f = f * absdot(ls.wi, sf.normal);
                const float pl = 1.0f / (float) sc.num_lights * ls.pdf;
                const float misWeight = powerHeuristic(1.0f, pl, 1.0f, pb);   // applied to delta lights too (Q10)
                ps.radiance = ps.radiance + ps.beta * (misWeight * f * ls.radiance / pl);
            } else {
                // sampleLights returns {} and integrateMIS still adds beta * {} (integrator.cpp:168,195):
                // a no-op for finite beta, NaN where the throughput has overflowed
                ps.radiance = ps.radiance + ps.beta * mk3(0.0f);
            }
        }
    }
    PH(3)
    const float u = ps.rng.f();
    f2 u2; u2.x = ps.rng.f(); u2.y = ps.rng.f();
    BSample bs;
    if (COUNT) { cnt.n_shade++; countClass(cnt.n_shade_t, bxdfClass(mat)); }
    if (!sampleBxdf<MASK>(ctx, mat, sf.normal, sf.uv, wo, u, u2, bs)) return BOUNCE_DONE;
    if (bs.pdf > 0.0f) ps.beta = ps.beta * (bs.f * absdot(bs.wi, sf.normal) / bs.pdf);
    ps.o = sf.point + bs.wi * RAY_EPSILON;                             // integrator.cpp:212
    ps.d = bs.wi;
    PH(4)
    return BOUNCE_NEXT;
}

JD void startPath(const DCam &cam, uint32_t row, uint32_t col, uint32_t s, PathState &ps) {
    ps.rng.seed(row, col, s + 1u);                                     // camera.cpp:101
    cameraRay(cam, col, row, s, ps.rng, ps.o, ps.d);
    ps.beta = mk3(1.0f); ps.radiance = mk3(0.0f); ps.depth = 0;
}

JD unsigned char toByte(float v) {                                     // image.hpp:9-16,47-52
    const float g = v > 0.0f ? sqrtf(v) : 0.0f;
    const float c = clampf(g, 0.0f, 0.999f);
    return (unsigned char) (int) (255.999f * c);
}

// SPLIT = strata-split mode: gridDim.y groups of strata per pixel block, per-sample radiance goes to p.rad and
// k_resolve_samples adds it to the film in sample order (same sums; more waves for small shards / frames).
// SRC: where the traversal reads the BVH -- SRC_GLOBAL threaded records in HBM, SRC_LDS the same staged in LDS,
// SRC_WIDE the 8-ary quantised nodes in HBM with the per-lane stack in LDS (uncounted kernels only).
template <bool COUNT, int SRC, int MASK, bool SPLIT>
__global__ void __launch_bounds__(BLOCK, SRC == SRC_WIDE ? JTX_WIDE_OCC : JTX_RP_OCC) k_render_pixels(RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    constexpr bool LDS_SCENE = SRC == SRC_LDS;
    const DevScene &sc = p.scene;
    float4 *lds_tnodes = (float4 *) smem;
    float4 *lds_tris = lds_tnodes + 2 * 8 * sc.num_nodes;
    if (LDS_SCENE) stageScene(sc, lds_tnodes, lds_tris);

    // work mapping: block = 4 waves = 4 consecutive 8x8 sub-blocks of one owned 32x32 tile
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bid = (int) blockIdx.x;
    const int owned = bid / BLOCKS_PER_TILE;                           // index into this rank's tiles
    const int tile = p.tile_rank + owned * p.tile_world;               // global 32x32 tile id, row-major (camera.cpp:55-64)
    const int sub = (bid % BLOCKS_PER_TILE) * WAVES_PER_BLOCK + wave;  // 0..15 sub-block inside the tile
    JTX_PROF_TIMELINE_BEGIN
    const int trow = tile / p.tiles_x, tcol = tile - trow * p.tiles_x;
    const int row = trow * 32 + (sub >> 2) * 8 + (lane >> 3);
    const int col = tcol * 32 + (sub & 3) * 8 + (lane & 7);
    const bool inside = row < p.height && col < p.width;

    Counters9 cnt = {};
    if (inside) {
        const size_t pix = (size_t) row * p.width + col;
        const int sBegin = SPLIT ? p.sample_begin + (int) blockIdx.y * p.strata_per_group : p.sample_begin;
        const int sEnd = SPLIT ? (sBegin + p.strata_per_group < p.sample_end ? sBegin + p.strata_per_group : p.sample_end) : p.sample_end;
        const int pslot = owned * 1024 + sub * 64 + lane;             // compact index of an owned pixel (strata-split mode)
        f3 acc = mk3(0.0f);
        if (!SPLIT && p.sample_begin > 0) acc = mk3(p.acc[3 * pix], p.acc[3 * pix + 1], p.acc[3 * pix + 2]);
        int s = sBegin;
        PathState ps;
        JTX_PROF_PHASES_BEGIN(ps)
        bool alive = s < sEnd;
        if (alive) { startPath(p.cam, row, col, s, ps); if (COUNT) cnt.n_camera++; }
        while (alive) {
            bool done;
            if constexpr (SRC == SRC_LDS) { LdsSrc src; src.tnodes = lds_tnodes; src.tris = lds_tris; src.half = 8 * sc.num_nodes; src.np = sc.num_prims;
                                  done = pathBounce<COUNT, MASK>(sc, src, p.max_depth, ps, cnt) == BOUNCE_DONE; }
            else if constexpr (SRC == SRC_WIDE) { WideSrc src; src.wide = sc.wide; src.tnodes = sc.tnodes; src.tris = sc.tris;
                                  src.stk = (uint2 *) smem + threadIdx.x; src.stride = BLOCK;
                                  done = pathBounce<COUNT, MASK>(sc, src, p.max_depth, ps, cnt) == BOUNCE_DONE; }
            else                { GlobalSrc src; src.tnodes = sc.tnodes; src.tris = sc.tris;
                                  done = pathBounce<COUNT, MASK>(sc, src, p.max_depth, ps, cnt) == BOUNCE_DONE; }
            if (done) {
                f3 c = ps.radiance;                                    // camera.cpp:110-112
                if (c.x > 1.0f) c.x = 1.0f;
                if (c.y > 1.0f) c.y = 1.0f;
                if (c.z > 1.0f) c.z = 1.0f;
                if (SPLIT) p.rad[(size_t) (s - p.sample_begin) * p.rad_stride + pslot] = make_float4(c.x, c.y, c.z, 0.0f);
                else acc = acc + c;                                    // image.hpp:82-86
                ++s;
                if (s < sEnd) { startPath(p.cam, row, col, s, ps); if (COUNT) cnt.n_camera++; }
                else alive = false;
            }
        }
        JTX_PROF_PHASES_END(p, ps, lane, false)
        if (!SPLIT) {
        p.acc[3 * pix] = acc.x; p.acc[3 * pix + 1] = acc.y; p.acc[3 * pix + 2] = acc.z;
        if (p.img) {
            const float inv = (float) p.sample_end;                    // currSample + 1 of the last pass (camera.cpp:115)
            p.img[3 * pix] = toByte(acc.x / inv);
            p.img[3 * pix + 1] = toByte(acc.y / inv);
            p.img[3 * pix + 2] = toByte(acc.z / inv);
        }
        }
    }
#ifdef JTX_PROFILE_TIMELINE
    if (!COUNT && p.counters && lane == 0 && p.sample_end - p.sample_begin > 1) {      // wave life span, indexed by pixel block
        const int wid = ((int) blockIdx.y * (int) gridDim.x + bid) * WAVES_PER_BLOCK + wave;
        JTX_PROF_TIMELINE_WAVE(p, wid)
    }
#endif
    if (COUNT) waveAddCounters(p.counters, cnt);
    if (SRC == SRC_WIDE) { JTX_PROF_WIDE_EXPORT(p, cnt) }
    if (COUNT) { JTX_PROF_UTIL_EXPORT(p, cnt) }
}

// ------------------------------------------------------------------------------------------------
// k_render_paths: DYNAMIC PATH ASSIGNMENT (uncounted kernels).
// In k_render_pixels a lane owns one pixel and a wave ends when its most expensive pixel does: wave timelines of the
// C2 launch show ~30 % of the lane time idle at the end of the waves (pixel costs differ 3x inside an 8x8 block), and
// halving the strata per lane raises the total wave time by 17 %.  Here the unit of work is ONE PATH: a wave holds
// a chunk = the strata-group x 64 paths of an 8x8 pixel block and hands them out from a wave-local counter (a ballot
// and a prefix count) -- stratum-major, so 64 lanes start with the 64 coherent camera rays of one stratum and a
// lane whose path ends takes the next path, whatever pixel it belongs to.  Every path writes its clamped
// radiance (camera.cpp:110-112) to rad[stratum][pixel] and k_resolve_samples adds them to the film in sample order,
// so the sums are those of AccumulationBuffer::updatePixel bit for bit.  Costs 32 B of HBM traffic per path.
// ------------------------------------------------------------------------------------------------
// BS = workgroup size: 256 when the scene is staged in LDS (one copy per 4 waves), 64 otherwise.
// The grid is PERSISTENT: as many workgroups as fit the GPU; a wave whose chunk -- one 8x8 pixel block x one strata
// group, 64 x strata paths, handed out stratum-major -- runs dry takes the next chunk from a global counter at once,
// while its other lanes are still finishing paths of the previous chunk: no wave ever drains except at the very end
// of the launch (a wave of 4 paths per lane lost ~12 % to its own drain), and the scene is staged once per workgroup.
// a finished path's clamped radiance (camera.cpp:110-112) into its record
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned PROG_NONE = 0xffffffffu;        // a wave's word of prog_slots: not started / gone; prog_closed_at: no cancellation
// WT: write-through (sc1) -- the record is read by ANOTHER kernel while this one still runs (k_resolve_progressive; per-XCD L2s are not
// coherent: a plain store would sit in this XCD's L2 until the launch ends)
template <bool WT>
JD void writeRadiance(const RenderParams &p, f3 c, int s, int slot) {
    if (c.x > 1.0f) c.x = 1.0f;
    if (c.y > 1.0f) c.y = 1.0f;
    if (c.z > 1.0f) c.z = 1.0f;
    if constexpr (WT) {
        u32x4 v; v.x = __float_as_uint(c.x); v.y = __float_as_uint(c.y); v.z = __float_as_uint(c.z); v.w = 0u;
        const float4 *q = p.rad + ((size_t) (s - p.sample_begin) * p.rad_stride + slot);
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(v) : "memory");
    } else
        p.rad[(size_t) (s - p.sample_begin) * p.rad_stride + slot] = make_float4(c.x, c.y, c.z, 0.0f);
}

// PROG (round 6): the PROGRESSIVE instance -- ONE launch for all passes of a frame (StaticCamera::render's loop over passes,
// camera.cpp:66-126, inside one kernel), beside k_resolve_progressive, which adds finished passes to the film while this kernel
// is still tracing later ones.  The path code is the same; what is added sits in the chunk fetch (once per 64 x strata paths):
// the wave publishes the OLDEST STRATUM it still has a path of (a wave-wide minimum over its live lanes and the chunk it has just
// fetched) after draining its record stores -- chunks are dealt in stratum order, so every record of every stratum below the minimum
// over all waves is in memory --, records are stored write-through, and a cancellation closes the chunk counter but keeps the
// chunk in hand (the chunks dealt so far are all traced: the resolver takes the passes they complete, an unfinished pass leaves
// no trace).  Costs the kernel 1.1 % (C2 22.68 -> 22.93 ms, C3 274.9 -> 278.1; the in-kernel commit it replaced: 17 - 33 %,
// EXPERIMENTS.md) -- the batch instance (PROG = false) is the code of round 5, instruction for instruction.
template <int SRC, int MASK, int BS, bool PROG>
__global__ void __launch_bounds__(BS, SRC == SRC_WIDE ? JTX_WIDE_OCC : JTX_RP_OCC) JTX_SGPR_ATTR k_render_paths(RenderParams p) {
    static_assert((SRC != SRC_LDS && SRC != SRC_LEAF) || BS == BLOCK, "stageScene strides by BLOCK");
    if constexpr (PROG) {
        // before the first fetch: "this wave may hold paths of any stratum" (a wave that has not come this far holds none: its word
        // stays PROG_NONE, and every chunk it will ever fetch lies behind the counter)
        if ((threadIdx.x & 63) == 0)
            __hip_atomic_store(p.work + PROG_CTL_SLOTS + ((int) blockIdx.x * (BS / 64) + (int) (threadIdx.x >> 6)), (unsigned) p.sample_begin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    extern __shared__ __attribute__((aligned(16))) int smem[];
    constexpr bool LDS_SCENE = SRC == SRC_LDS || SRC == SRC_LEAF;
    const DevScene &sc = p.scene;
    float4 *lds_tnodes = (float4 *) smem;
    float4 *lds_tris = lds_tnodes + 2 * 8 * sc.num_nodes;
    float4 *lds_lbox = lds_tris + 3 * sc.num_prims;                     // SRC_LEAF: [leaf list][order / position tables]
    const int lwPad = (sc.lw_leaves + 3) & ~3;
    unsigned *lds_tab = (unsigned *) (lds_lbox + 2 * lwPad);
    if (SRC == SRC_LEAF) {
#if JTX_LDS_PLANES
        for (int i = threadIdx.x; i < 2 * lwPad; i += BS) lds_lbox[(i & 1) * lwPad + (i >> 1)] = sc.lw_box[i];   // halves apart
#else
        for (int i = threadIdx.x; i < 2 * lwPad; i += BS) lds_lbox[i] = sc.lw_box[i];
#endif
        for (int i = threadIdx.x; i < 128; i += BS) lds_tab[i] = sc.lw_tab[i];
    }
    if (LDS_SCENE) stageScene(sc, lds_tnodes, lds_tris);
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int nchunks = p.num_subblocks * p.num_groups;
    JTX_PROF_TIMELINE_BEGIN

    Counters9 cnt = {};
    PathState ps;
    JTX_PROF_PHASES_BEGIN(ps)
    // the wave's current chunk (all wave-uniform)
    int next = 0, nunits = 0;                      // paths handed out / in the chunk
    int row0 = 0, col0 = 0, slot0 = 0, sBegin = 0;
    bool exhausted = false;
    int s = 0, slot = 0;                           // this lane's current path: stratum, pixel slot in the rank's frame
    bool alive = false, need = true;
#ifdef JTX_DBG_PROG
    int dbgFetches = 0;
#endif
    while (true) {
        JTX_PROF_HANDOUT_BEGIN
        // ---- hand out paths to the lanes that need one ----
        while (true) {
            const unsigned long long mask = __ballot(need);
            if (mask == 0ull) break;
            if (next >= nunits) {                                        // chunk used up: fetch the next one
                if (exhausted) { need = false; break; }
                int c = 0;
#ifdef JTX_NO_STOP_POLL            /* A/B only: what the cancellation poll costs */
                if (lane == 0) c = (int) atomicAdd(p.work, 1u);
#else
                if (lane == 0) {
                    c = (int) atomicAdd(p.work, 1u);
                    // cancellation poll: the flag lives in HOST memory (one PCIe read), so only every 64th fetch looks, and the
                    // wave that sees it pushes the chunk counter past the end: every other wave stops at its next fetch
                    if (p.stop && (c & 63) == 0 && c < nchunks && __hip_atomic_load(p.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
                        if constexpr (PROG) {                            // the chunk in hand is traced: chunks [0, old) are all of the launch
                            const unsigned old = atomicMax(p.work, 0x40000000u);
                            if (old < 0x40000000u) __hip_atomic_store(p.work + PROG_CTL_CLOSED_AT, old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else { atomicMax(p.work, 0x40000000u); c = nchunks; }
                    }
                }
#endif
                c = __shfl(c, 0, 64);
#ifdef JTX_DBG_PROG      /* diagnostic build: this wave's fetches so far and where it runs, beside its word (read by the resolver leader's snapshot) */
                if constexpr (PROG) {
                    ++dbgFetches;
                    if (lane == 0) { const int w = (int) blockIdx.x * (BS / 64) + (int) (threadIdx.x >> 6);
                                     __hip_atomic_store(p.work + PROG_CTL_SLOTS + 16384 + w, (unsigned) dbgFetches, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                     if (dbgFetches == 1) __hip_atomic_store(p.work + PROG_CTL_SLOTS + 32768 + w, (unsigned) __builtin_amdgcn_s_getreg(63492), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                }
#endif
                if constexpr (PROG) {
                    int m = alive ? s : 0x7fffffff;                      // the oldest stratum this wave still has a path of ...
                    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(m, off, 64); m = o < m ? o : m; }
                    const int first = c < nchunks ? p.sample_begin + (c / p.num_subblocks) * p.strata_per_group : 0x7fffffff;
                    m = first < m ? first : m;                           // ... or is about to start
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the records of the paths that have ended are out
                    if (lane == 0) __hip_atomic_store(p.work + PROG_CTL_SLOTS + ((int) blockIdx.x * (BS / 64) + (int) (threadIdx.x >> 6)),
                                                      m == 0x7fffffff ? PROG_NONE : (unsigned) m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (c >= nchunks) { exhausted = true; need = false; break; }
                // chunk c: strata group-major, so that the whole frame advances stratum range by stratum range
                const int grp = c / p.num_subblocks, sb8 = c - grp * p.num_subblocks;      // 8x8 block index of this rank
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
                    startPath(p.cam, row, col, s, ps);
                    alive = true; need = false;
                }
            }
        }
        need = false;
        JTX_PROF_HANDOUT_END(ps)
        if (__ballot(alive) == 0ull) break;
        JTX_PROF_TIMELINE_ITER(alive)
        // ---- one bounce of every live path ----
        if (alive) {
            bool done;
            if constexpr (SRC == SRC_LDS) { LdsSrc src; src.tnodes = lds_tnodes; src.tris = lds_tris; src.half = 8 * sc.num_nodes; src.np = sc.num_prims;
                                  done = pathBounce<false, MASK>(sc, src, p.max_depth, ps, cnt) == BOUNCE_DONE; }
            else if constexpr (SRC == SRC_LEAF) { LeafSrc src; src.tnodes = lds_tnodes; src.tris = lds_tris; src.half = 8 * sc.num_nodes;
                                  src.lbox = lds_lbox; src.gbox = sc.lw_box; src.groot = sc.tnodes; src.fresh = ps.depth == 0; src.tab = lds_tab; src.nleaf = sc.lw_leaves; src.np = sc.num_prims; src.lpad = lwPad;
                                  done = pathBounce<false, MASK>(sc, src, p.max_depth, ps, cnt) == BOUNCE_DONE; }
            else if constexpr (SRC == SRC_WIDE) { WideSrcT<(MASK == MAT_DIFFUSE_ONLY ? JTX_WIDE_STEPS_LC : JTX_WIDE_STEPS_MC), (MASK == MAT_DIFFUSE_ONLY ? JTX_WIDE_STEPS_LA : JTX_WIDE_STEPS_MA)> src;
                                  src.wide = sc.wide; src.tnodes = sc.tnodes; src.tris = sc.tris;
                                  src.stk = (uint2 *) smem + threadIdx.x; src.stride = BS;
                                  done = pathBounce<false, MASK>(sc, src, p.max_depth, ps, cnt) == BOUNCE_DONE; }
            else                { GlobalSrc src; src.tnodes = sc.tnodes; src.tris = sc.tris;
                                  done = pathBounce<false, MASK>(sc, src, p.max_depth, ps, cnt) == BOUNCE_DONE; }
            if (done) {
                writeRadiance<PROG>(p, ps.radiance, s, slot);
                alive = false; need = true;
            }
        }
    }
    if constexpr (PROG) {                                               // gone: every record of this wave is out
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(p.work + PROG_CTL_SLOTS + ((int) blockIdx.x * (BS / 64) + (int) (threadIdx.x >> 6)), PROG_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (SRC == SRC_WIDE) { JTX_PROF_WIDE_EXPORT(p, cnt) }
    JTX_PROF_PHASES_END(p, ps, lane, true)
    JTX_PROF_TIMELINE_END(p, lane, (int) blockIdx.x * (BS / 64) + (int) (threadIdx.x >> 6))
}

// ------------------------------------------------------------------------------------------------
// parity-test kernels
// ------------------------------------------------------------------------------------------------
// Per-ray entry points, instantiated for every BVH source the render kernels walk (jtx_mi_closest_hit_batch_via / any_hit_batch_via):
// SRC_GLOBAL the binary threaded records in HBM (what the counted kernels walk), SRC_LDS the same staged in LDS, SRC_LEAF the flat leaf
// list (traverseLeaves: scalar-operand boxes, candidates from the LDS copy), SRC_WIDE the 8-ary quantised nodes with the per-lane LDS
// stack (traverseWide) -- the SAME Src objects, built the same way, as k_render_paths builds them, so a ray through these kernels runs
// the code a path's ray runs in the timed launch (incl. the wave-wide vote that sends a wave with an irregular ray to the binary records).
template <int SRC, int BS, class F>
JD void withBatchSrc(const DevScene &sc, int *smem, F &&body) {
    static_assert((SRC != SRC_LDS && SRC != SRC_LEAF) || BS == BLOCK, "stageScene strides by BLOCK");
    constexpr bool LDS_SCENE = SRC == SRC_LDS || SRC == SRC_LEAF;
    float4 *lds_tnodes = (float4 *) smem;
    float4 *lds_tris = lds_tnodes + 2 * 8 * sc.num_nodes;
    float4 *lds_lbox = lds_tris + 3 * sc.num_prims;                     // SRC_LEAF: [leaf list][order / position tables]
    const int lwPad = (sc.lw_leaves + 3) & ~3;
    unsigned *lds_tab = (unsigned *) (lds_lbox + 2 * lwPad);
    if (SRC == SRC_LEAF) {
#if JTX_LDS_PLANES
        for (int i = threadIdx.x; i < 2 * lwPad; i += BS) lds_lbox[(i & 1) * lwPad + (i >> 1)] = sc.lw_box[i];   // halves apart
#else
        for (int i = threadIdx.x; i < 2 * lwPad; i += BS) lds_lbox[i] = sc.lw_box[i];
#endif
        for (int i = threadIdx.x; i < 128; i += BS) lds_tab[i] = sc.lw_tab[i];
    }
    if (LDS_SCENE) stageScene(sc, lds_tnodes, lds_tris);                // (every thread of the workgroup: before any lane leaves)
    if constexpr (SRC == SRC_LDS) { LdsSrc src; src.tnodes = lds_tnodes; src.tris = lds_tris; src.half = 8 * sc.num_nodes; src.np = sc.num_prims; body(src); }
    else if constexpr (SRC == SRC_LEAF) { LeafSrc src; src.tnodes = lds_tnodes; src.tris = lds_tris; src.half = 8 * sc.num_nodes;
                          src.lbox = lds_lbox; src.gbox = sc.lw_box; src.tab = lds_tab; src.nleaf = sc.lw_leaves; src.np = sc.num_prims; src.lpad = lwPad;
                          src.groot = sc.tnodes; src.fresh = true;        // the per-ray entry points take the root test of the path kernel's camera rays too
                          body(src); }
    else if constexpr (SRC == SRC_WIDE) { WideSrc src; src.wide = sc.wide; src.tnodes = sc.tnodes; src.tris = sc.tris;
                          src.stk = (uint2 *) smem + threadIdx.x; src.stride = BS; body(src); }
    else                { GlobalSrc src; src.tnodes = sc.tnodes; src.tris = sc.tris; body(src); }
}

template <int SRC, int BS>
__global__ void __launch_bounds__(BS) k_closest_batch(DevScene sc, int n, const float *o, const float *d, float tmin,
                                                      float tmax, int *hit, float *t, int *prim, float *b1, float *b2,
                                                      float *point, float *normal, float *uv) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    withBatchSrc<SRC, BS>(sc, smem, [&](const auto &src) {
        const int i = blockIdx.x * BS + threadIdx.x;
        if (i >= n) return;
        Counters9 cnt = {};
        HitRec h; h.t = 0.0f; h.prim = -1; h.b1 = h.b2 = 0.0f;
        const f3 ro = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        const bool r = traverseNoStack<false, false>(src, sc.num_nodes, ro, rd, tmin, tmax, h, cnt);
        hit[i] = r ? 1 : 0;
        Surface sf; sf.point = sf.normal = mk3(0.0f); sf.uv = mk2(0.0f, 0.0f);
        if (r) sf = makeSurface(sc.shade, h, ro, rd); else { h.t = 0.0f; h.prim = -1; h.b1 = h.b2 = 0.0f; }
        t[i] = h.t; prim[i] = h.prim; b1[i] = h.b1; b2[i] = h.b2;
        point[3 * i] = sf.point.x; point[3 * i + 1] = sf.point.y; point[3 * i + 2] = sf.point.z;
        normal[3 * i] = sf.normal.x; normal[3 * i + 1] = sf.normal.y; normal[3 * i + 2] = sf.normal.z;
        uv[2 * i] = sf.uv.x; uv[2 * i + 1] = sf.uv.y;
    });
}

template <int SRC, int BS>
__global__ void __launch_bounds__(BS) k_any_batch(DevScene sc, int n, const float *o, const float *d, const float *tmin,
                                                  const float *tmax, int *hit) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    withBatchSrc<SRC, BS>(sc, smem, [&](const auto &src) {
        const int i = blockIdx.x * BS + threadIdx.x;
        if (i >= n) return;
        Counters9 cnt = {};
        HitRec h;
        const f3 ro = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        hit[i] = traverseNoStack<true, false>(src, sc.num_nodes, ro, rd, tmin[i], tmax[i], h, cnt) ? 1 : 0;
    });
}

__global__ void __launch_bounds__(BLOCK) k_bxdf_batch(DevScene sc, int mode, int material, int n, const float *normal,
                                                      const float *uv, const float *wo, const float *wi_in, const float *uc,
                                                      const float *u2, int *ok, float *f, float *wi_out, float *pdf) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    ShadeCtx ctx; ctx.materials = sc.materials; ctx.textures = sc.textures; ctx.texels = sc.texels;
    const DMaterial &m = sc.materials[material];
    const f3 nrm = mk3(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
    const f2 tuv = uv ? mk2(uv[2 * i], uv[2 * i + 1]) : mk2(0.0f, 0.0f);
    const f3 w_o = mk3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]);
    if (mode == 0) {
        BSample bs; bs.f = bs.wi = mk3(0.0f); bs.pdf = 0.0f;
        const bool r = sampleBxdf<MAT_EVERY>(ctx, m, nrm, tuv, w_o, uc[i], mk2(u2[2 * i], u2[2 * i + 1]), bs);
        if (!r) { bs.f = bs.wi = mk3(0.0f); bs.pdf = 0.0f; }
        ok[i] = r ? 1 : 0;
        f[3 * i] = bs.f.x; f[3 * i + 1] = bs.f.y; f[3 * i + 2] = bs.f.z;
        wi_out[3 * i] = bs.wi.x; wi_out[3 * i + 1] = bs.wi.y; wi_out[3 * i + 2] = bs.wi.z;
        pdf[i] = bs.pdf;
    } else {
        const f3 w_i = mk3(wi_in[3 * i], wi_in[3 * i + 1], wi_in[3 * i + 2]);
        f3 fv; float pv;
        evalPdfBxdf<MAT_EVERY>(ctx, m, nrm, tuv, w_o, w_i, fv, pv);
        if (f) { f[3 * i] = fv.x; f[3 * i + 1] = fv.y; f[3 * i + 2] = fv.z; }
        if (pdf) pdf[i] = pv;
    }
}

__global__ void __launch_bounds__(BLOCK) k_camera_rays(DCam cam, int n, const int *row, const int *col, const int *sample,
                                                       float *o, float *d) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    Rng rng; rng.seed(row[i], col[i], sample[i] + 1);
    f3 ro, rd; cameraRay(cam, col[i], row[i], sample[i], rng, ro, rd);
    o[3 * i] = ro.x; o[3 * i + 1] = ro.y; o[3 * i + 2] = ro.z;
    d[3 * i] = rd.x; d[3 * i + 1] = rd.y; d[3 * i + 2] = rd.z;
}

__global__ void __launch_bounds__(BLOCK) k_radiance_samples(DevScene sc, DCam cam, int maxDepth, int n, const int *row,
                                                            const int *col, const int *sample, float *rgb) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    Counters9 cnt = {};
    PathState ps;
    startPath(cam, row[i], col[i], sample[i], ps);
    GlobalSrc src; src.tnodes = sc.tnodes; src.tris = sc.tris;
    while (pathBounce<false, MAT_ALL>(sc, src, maxDepth, ps, cnt) != BOUNCE_DONE) {}
    f3 c = ps.radiance;
    if (c.x > 1.0f) c.x = 1.0f;
    if (c.y > 1.0f) c.y = 1.0f;
    if (c.z > 1.0f) c.z = 1.0f;
    rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
}

__global__ void k_rng_stream(uint32_t x, uint32_t y, uint32_t n, int count, uint32_t *out_u32, float *out_f32) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Rng a, b; a.seed(x, y, n); b.seed(x, y, n);
    for (int i = 0; i < count; ++i) { out_u32[i] = a.advance(); out_f32[i] = b.f(); }
}

__global__ void __launch_bounds__(BLOCK) k_sincos(const float *x, int n, float *s, float *c) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float ss, cc; det_sincos(x[i], ss, cc);
    s[i] = ss; c[i] = cc;
}

} // namespace jtx

// ------------------------------------------------------------------------------------------------
// launchers (host)
// ------------------------------------------------------------------------------------------------
using namespace jtx;

static size_t ldsBytes(const DevScene &sc, bool withScene) {      // stackless: only the staged scene, if any
    size_t b = 0;
    if (withScene) b += ((size_t) 2 * 8 * sc.num_nodes + (size_t) 3 * sc.num_prims) * sizeof(float4);
    return b;
}

hipError_t jtx_launch_render_pixels(const RenderParams &p, int num_owned_tiles, bool count, hipStream_t stream) {
    if (num_owned_tiles <= 0) return hipSuccess;
    const dim3 grid((unsigned) num_owned_tiles * (unsigned) BLOCKS_PER_TILE), block(BLOCK);
    const bool lds = p.scene.lds_threaded != 0;
    const bool wide = !lds && !count && p.scene.wide != nullptr;
    const size_t shmem = wide ? (size_t) p.scene.wide_depth * BLOCK * sizeof(uint2) : ldsBytes(p.scene, lds);
    const bool lambert = p.scene.material_mask == MAT_DIFFUSE_ONLY;
    const bool split = p.rad != nullptr;
    const int groups = split ? (p.sample_end - p.sample_begin + p.strata_per_group - 1) / p.strata_per_group : 1;
    const dim3 grid2(grid.x, (unsigned) groups);
#define LAUNCH_RP(C, L, M) do { if (split) hipLaunchKernelGGL((k_render_pixels<C, L, M, true>), grid2, block, shmem, stream, p); \
                                else hipLaunchKernelGGL((k_render_pixels<C, L, M, false>), grid, block, shmem, stream, p); } while (0)
    if (lambert) {
        if (lds)       { if (count) LAUNCH_RP(true, SRC_LDS, MAT_DIFFUSE_ONLY); else LAUNCH_RP(false, SRC_LDS, MAT_DIFFUSE_ONLY); }
        else if (wide) LAUNCH_RP(false, SRC_WIDE, MAT_DIFFUSE_ONLY);
        else           { if (count) LAUNCH_RP(true, SRC_GLOBAL, MAT_DIFFUSE_ONLY); else LAUNCH_RP(false, SRC_GLOBAL, MAT_DIFFUSE_ONLY); }
    } else {
        if (lds)       { if (count) LAUNCH_RP(true, SRC_LDS, MAT_ALL); else LAUNCH_RP(false, SRC_LDS, MAT_ALL); }
        else if (wide) LAUNCH_RP(false, SRC_WIDE, MAT_ALL);
        else           { if (count) LAUNCH_RP(true, SRC_GLOBAL, MAT_ALL); else LAUNCH_RP(false, SRC_GLOBAL, MAT_ALL); }
    }
#undef LAUNCH_RP
    return hipGetLastError();
}

// strata-split mode: add the per-sample radiances to the film in sample order (image.hpp:82-86, 47-52)
namespace jtx {
__global__ void __launch_bounds__(BLOCK) k_resolve_samples(RenderParams p) {
    const int pslot = blockIdx.x * BLOCK + threadIdx.x;
    const int owned = pslot >> 10, sub = (pslot >> 6) & 15, lane = pslot & 63;
    const int tile = p.tile_rank + owned * p.tile_world;
    const int trow = tile / p.tiles_x, tcol = tile - trow * p.tiles_x;
    const int row = trow * 32 + (sub >> 2) * 8 + (lane >> 3);
    const int col = tcol * 32 + (sub & 3) * 8 + (lane & 7);
    if (pslot >= p.rad_stride || row >= p.height || col >= p.width) return;
    // a cancelled pass has incomplete records: the wave that saw the cancellation pushed the chunk counter past 2^30
    // (device memory: one cached load per thread, not one PCIe read of the host's flag)
    const bool own = p.work && *(volatile const unsigned *) p.work >= 0x40000000u;
    // ... or the pass before this one was: then this one does not enter the film either, and says so to the next
    const bool prev = p.prev_work && *(volatile const unsigned *) p.prev_work >= 0x40000000u;
    if (own || prev) {
        if (pslot == 0) {
            if (prev && !own) atomicMax(p.work, 0x40000000u);
            if (p.abandoned) __hip_atomic_store(p.abandoned, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // the host's copy of the verdict
        }
        return;
    }
    const size_t pix = (size_t) row * p.width + col;
    f3 acc = mk3(0.0f);
    if (p.sample_begin > 0) acc = mk3(p.acc[3 * pix], p.acc[3 * pix + 1], p.acc[3 * pix + 2]);
    const int n = p.sample_end - p.sample_begin;
    for (int i = 0; i < n; ++i) {
        const float4 c = p.rad[(size_t) i * p.rad_stride + pslot];
        acc = acc + mk3(c.x, c.y, c.z);
    }
    p.acc[3 * pix] = acc.x; p.acc[3 * pix + 1] = acc.y; p.acc[3 * pix + 2] = acc.z;
    if (p.img) {
        const float inv = (float) p.sample_end;
        p.img[3 * pix] = toByte(acc.x / inv); p.img[3 * pix + 1] = toByte(acc.y / inv); p.img[3 * pix + 2] = toByte(acc.z / inv);
    }
}
} // namespace jtx

int jtx_render_paths_grid(const DevScene &sc, int num_cus, int *block_size) {
    const bool lds = sc.lds_threaded != 0, wide = !lds && sc.wide != nullptr;
    const int bs = lds ? BLOCK : 64;
    if (block_size) *block_size = bs;
    return (int) (((long) num_cus * 4 * (wide ? JTX_WIDE_OCC : JTX_RP_OCC) * 64 + bs - 1) / bs);
}

// share: this launch takes 1 / share of the wave slots (several small launches that fill the chip together)
// leave_waves: wave slots left to a kernel that runs beside it (the progressive resolver)
constexpr int RESOLVE_BLOCK_HOST = 256;            // = jtx::RESOLVE_BLOCK (k_resolve_progressive)
static long renderPathsWaves(const RenderParams &p, int num_cus, int share, int leave_waves, int *bsOut) {
    const bool lds = p.scene.lds_threaded != 0;
    const bool wide = !lds && p.scene.wide != nullptr;
    const int bs = lds ? BLOCK : 64;                                     // 64: the workgroup of the kernels that stage nothing
    if (bsOut) *bsOut = bs;
    // persistent grid: the waves the GPU can hold (occupancy of the launch bounds), no more than there are chunks
    const int occ = wide ? JTX_WIDE_OCC : JTX_RP_OCC;
    long waves = (long) num_cus * 4 * occ;
    if (share > 1) waves = (waves / share + (bs / 64) - 1) / (bs / 64) * (bs / 64);
    if (leave_waves > 0 && waves > (long) 2 * leave_waves) waves -= ((long) leave_waves + (bs / 64) - 1) / (bs / 64) * (bs / 64);
    const long chunks = (long) p.num_subblocks * p.num_groups;
    if (waves > chunks) waves = chunks;
    return ((waves * 64 + bs - 1) / bs) * (bs / 64);                     // whole workgroups
}
int jtx_render_paths_waves(const RenderParams &p, int num_cus, int share, int leave_waves) { return (int) renderPathsWaves(p, num_cus, share, leave_waves, nullptr); }
int jtx_resolve_progressive_waves(int num_workgroups) { return num_workgroups * (RESOLVE_BLOCK_HOST / 64); }

hipError_t jtx_launch_render_paths(const RenderParams &p, int num_owned_tiles, int num_cus, hipStream_t stream, int share, bool progressive, int leave_waves) {
    if (num_owned_tiles <= 0) return hipSuccess;
    const bool lds = p.scene.lds_threaded != 0;
    const bool wide = !lds && p.scene.wide != nullptr;
    constexpr int SMALL = 64;                                            // workgroup of the kernels that stage nothing
    int bs = 0;
    const long waves = renderPathsWaves(p, num_cus, share, leave_waves, &bs);
    static const int leafWalk = [] { const char *e = getenv("JTX_LEAF_WALK"); return e ? atoi(e) : 1; }();
    const bool leaf = lds && p.scene.lw_leaves > 0 && leafWalk;
    const size_t shmem = wide ? (size_t) p.scene.wide_depth * bs * sizeof(uint2)
                              : ldsBytes(p.scene, lds) + (leaf ? (size_t) 2 * ((p.scene.lw_leaves + 3) & ~3) * sizeof(float4) + 128 * sizeof(unsigned) : 0);
    const bool lambert = p.scene.material_mask == MAT_DIFFUSE_ONLY;
    const dim3 grid((unsigned) (waves * 64 / bs)), block(bs);
#define LAUNCH_PA(L, M, B) do { if (progressive) hipLaunchKernelGGL((k_render_paths<L, M, B, true>), grid, block, shmem, stream, p); \
                                else hipLaunchKernelGGL((k_render_paths<L, M, B, false>), grid, block, shmem, stream, p); } while (0)
    if (leaf) { if (lambert) LAUNCH_PA(SRC_LEAF, MAT_DIFFUSE_ONLY, BLOCK); else LAUNCH_PA(SRC_LEAF, MAT_ALL, BLOCK); }
    else if (lambert) { if (lds) LAUNCH_PA(SRC_LDS, MAT_DIFFUSE_ONLY, BLOCK); else if (wide) LAUNCH_PA(SRC_WIDE, MAT_DIFFUSE_ONLY, SMALL); else LAUNCH_PA(SRC_GLOBAL, MAT_DIFFUSE_ONLY, SMALL); }
    else         { if (lds) LAUNCH_PA(SRC_LDS, MAT_ALL, BLOCK); else if (wide) LAUNCH_PA(SRC_WIDE, MAT_ALL, SMALL); else LAUNCH_PA(SRC_GLOBAL, MAT_ALL, SMALL); }
#undef LAUNCH_PA
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// k_resolve_progressive: the film side of a progressive launch.  A few persistent workgroups beside k_render_paths<.., PROG>: each owns a
// fixed share of the rank's pixels and adds, pass after pass, the records of every pass that is COMPLETE -- in sample order per pixel,
// AccumulationBuffer::updatePixel's float sums (image.hpp:82-86) -- and writes the RGB8 preview (image.hpp:47-52; the divisor is the
// strata so far, camera.cpp:115).  Complete = dealt and out of every wave's hands:
//     dealt  = chunks the counter has handed out (after a cancellation: the count it was closed at) -- read FIRST;
//     oldest = the minimum over the path waves' words of prog_slots (the oldest stratum a wave still has a path of; a wave stores the
//              launch's first stratum before its first fetch and drains its record stores before every update: a stale word is a
//              smaller one) -- read SECOND, so that every chunk counted in `dealt` is covered by its wave's word;
//     passes complete = min(dealt / chunks per pass, (oldest - first stratum) / strata per pass).
// A pass that a cancellation left unfinished is never added: the film then holds exactly the passes before it.
// Workgroup 0 watches the path waves and publishes the count (one workgroup scanning: 64 of them polling the chunk counter and 7 000 words
// each slowed the path kernel from 25.8 to 33.5 ms); the others add the passes, each for a fixed share of the pixels.  The path kernel
// never waits for this one: if the resolver finds no wave slots until the path kernel has ended, it adds all passes then (what
// k_resolve_samples does for a batch launch).
// progress_host[workgroup] = epoch << 16 | passes in the film of ALL its pixels (host-mapped; the host takes the minimum: currentSample_).
// ------------------------------------------------------------------------------------------------
namespace jtx {
constexpr int RESOLVE_BLOCK = 256;
constexpr unsigned RESOLVE_LAST = 0x80000000u;      // the leader's word: bit 31 = this count is final (every chunk dealt, every wave gone)
constexpr unsigned RESOLVE_GAVE_UP = 0x40000000u;   // ... bit 30 = the leader gave up: nothing moved for the launch's patience (the path kernel never came)
constexpr unsigned RESOLVE_PATIENCE_MS = 60000u;   // a minute without a chunk fetched or a wave's word moving (the launcher may pass another)

// pixel slot -> its pixel; false for the padding slots of tiles that overhang the frame
JD bool resolveSlot(const RenderParams &p, int pslot, size_t &pix) {
    if (pslot >= p.rad_stride) return false;
    const int owned = pslot >> 10, sub = (pslot >> 6) & 15, l = pslot & 63;
    const int tile = p.tile_rank + owned * p.tile_world;
    const int trow = tile / p.tiles_x, tcol = tile - trow * p.tiles_x;
    const int row = trow * 32 + (sub >> 2) * 8 + (l >> 3);
    const int col = tcol * 32 + (sub & 3) * 8 + (l & 7);
    pix = (size_t) row * p.width + col;
    return row < p.height && col < p.width;
}
JD void resolveStore(const RenderParams &p, size_t pix, f3 acc, int sB) {
    p.acc[3 * pix] = acc.x; p.acc[3 * pix + 1] = acc.y; p.acc[3 * pix + 2] = acc.z;
    if (p.img) {                                                         // write-through: the host copies previews while the launch runs
        const float inv = (float) sB;                                    // currSample + 1 of the last pass added (camera.cpp:115)
        unsigned char *q = p.img + 3 * pix;
        __hip_atomic_store(q, toByte(acc.x / inv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, toByte(acc.y / inv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 2, toByte(acc.z / inv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void __launch_bounds__(RESOLVE_BLOCK) k_resolve_progressive(RenderParams p, int num_path_waves, unsigned *started_host, unsigned *progress_host, unsigned *keepalive_host, unsigned epoch,
                                                                          unsigned patience_ms) {
    __shared__ unsigned shWord, shGive, shOldest[RESOLVE_BLOCK / 64];
    const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
    if (tid == 0) __hip_atomic_store(started_host + wg, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int nchunks = p.num_subblocks * p.num_groups;
    unsigned *leaderWord = p.prog_leader;
    if (wg == 0 && nwg > 1) {
        // ---- the leader: how many passes are complete?  (One workgroup watches the path waves, the others watch its word.) ----
        unsigned dealtSeen = 0u, published = 0u, sigBefore = 0xffffffffu;
#ifdef JTX_DBG_PROG
        bool dbgSnap = false;
#endif
        unsigned long long tMoved = 0ull;                             // (thread 0's: when something last moved)
        if (tid == 0) __hip_atomic_store(progress_host + wg, (epoch << 16) | (unsigned) p.num_groups, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (no pixels of its own: never the minimum)
        while (true) {
            if (tid == 0) {
                const unsigned wk = atomicAdd(p.work, 0u);               // (one workgroup, once a round: nothing beside the waves' fetches)
                unsigned dealt, known = RESOLVE_LAST;                    // bit 31: `dealt` is final once every wave is gone
                if (wk < 0x40000000u) { dealt = wk < (unsigned) nchunks ? wk : (unsigned) nchunks; if (wk < (unsigned) nchunks) known = 0u; }
                else {                                                  // closed by a cancellation: the wave that closed it says at what count
                    const unsigned ca = atomicOr(p.prog_closed_at, 0u);
                    if (ca == PROG_NONE) { dealt = dealtSeen; known = 0u; } else dealt = ca < (unsigned) nchunks ? ca : (unsigned) nchunks;
                }
                shWord = dealt | known;
            }
            __syncthreads();
            const unsigned dealt = shWord & ~RESOLVE_LAST;
            const bool dealtFinal = (shWord & RESOLVE_LAST) != 0u;
            dealtSeen = dealt;
            unsigned m = PROG_NONE;
            // the words as they are in memory NOW: atomics.  (sc1 loads were served from this XCD's L2 for ~9 ms after a wave on another XCD
            // had rewritten a word -- a stale word is a smaller one, so nothing was wrong, but every preview came three passes late.)
            for (int i = tid; i < num_path_waves; i += RESOLVE_BLOCK) { const unsigned v = atomicOr(p.prog_slots + i, 0u); m = v < m ? v : m; }
            for (int off = 32; off > 0; off >>= 1) { const unsigned o = __shfl_xor(m, off, 64); m = o < m ? o : m; }
            if ((tid & 63) == 0) shOldest[tid >> 6] = m;
            __syncthreads();
            unsigned oldest = shOldest[0];
            for (int i = 1; i < RESOLVE_BLOCK / 64; ++i) oldest = shOldest[i] < oldest ? shOldest[i] : oldest;
            __syncthreads();                                            // (shWord / shOldest are rewritten in the next round)
#ifdef JTX_DBG_PROG      /* once, when two groups have been dealt: which waves still hold a path of group 0, how many chunks they and the others have fetched */
            if (!dbgSnap && dealt >= 2u * (unsigned) p.num_subblocks) {
                dbgSnap = true;
                unsigned *dg = p.prog_slots + 49152;                     // [0] laggards [1] sum of their fetches [2] min fetches of all [3] max [4] sum of all [5] waves seen [6] entries, then (wave, fetches, hw id, word) x 16
                if (tid < 48) dg[80 + tid] = 0u;
                if (tid == 0) { dg[0] = dg[1] = dg[3] = dg[4] = dg[5] = dg[6] = 0u; dg[2] = 0xffffffffu; dg[7] = (unsigned) (__builtin_amdgcn_s_memrealtime() / 100); }
                __threadfence(); __syncthreads();
                for (int i = tid; i < num_path_waves; i += RESOLVE_BLOCK) {
                    const unsigned v = atomicOr(p.prog_slots + i, 0u), f = atomicOr(p.prog_slots + 16384 + i, 0u), hw = atomicOr(p.prog_slots + 32768 + i, 0u);
                    if (v == PROG_NONE && f == 0xffffffffu) continue;   // (never started)
                    atomicAdd(dg + 5, 1u); atomicAdd(dg + 4, f); atomicMin(dg + 2, f); atomicMax(dg + 3, f);
                    atomicAdd(dg + 80 + (hw & 15u), 1u); atomicAdd(dg + 96 + (hw & 15u), f);           // by the wave's slot in its SIMD
                    if (v != PROG_NONE && (int) v < p.sample_begin + p.strata_per_group) atomicAdd(dg + 112 + (hw & 15u), 1u);
                    if (v != PROG_NONE && (int) v < p.sample_begin + p.strata_per_group) {
                        atomicAdd(dg + 0, 1u); atomicAdd(dg + 1, f);
                        const unsigned e = atomicAdd(dg + 6, 1u);
                        if (e < 16u) { dg[8 + 4 * e] = (unsigned) i; dg[9 + 4 * e] = f; dg[10 + 4 * e] = hw; dg[11 + 4 * e] = v; }
                    }
                }
                __syncthreads();
            }
#endif
            const bool allGone = oldest == PROG_NONE;
            int complete = (int) (dealt / (unsigned) p.num_subblocks);
            if (!allGone) { const int byWaves = (int) oldest > p.sample_begin ? ((int) oldest - p.sample_begin) / p.strata_per_group : 0; complete = byWaves < complete ? byWaves : complete; }
            if (complete >= p.num_groups) complete = p.num_groups;
            else complete = complete / p.prog_groups_per_pass * p.prog_groups_per_pass;    // whole passes only (a cancellation may end the launch inside one)
            bool last = allGone && dealtFinal;
            // a bounded wait: should nothing move for a minute -- no chunk fetched, no wave's word changed -- AND the host no longer vouch for
            // the path kernel, the resolver ends, with the passes it has, and says so.  The host vouches (a word in host-mapped memory, rewritten
            // by its polling loop) as long as the path kernel has not ended: a path kernel that waits its turn behind another process' or
            // another scene's long launch is no reason to give up.  ONE thread reads the clock (the waves of a workgroup read it at different
            // moments: each deciding for itself could split the workgroup at its barrier).
            const unsigned sig = shWord ^ (oldest * 0x9e3779b9u);
            if (sig != sigBefore) { sigBefore = sig; tMoved = 0ull; }        // (0: take the time at the next look)
            if (tid == 0) {
                const unsigned long long tNow = __builtin_amdgcn_s_memrealtime();
                if (tMoved == 0ull) tMoved = tNow;
                bool give = !last && tNow - tMoved > (unsigned long long) patience_ms * 100000ull;      // s_memrealtime ticks: 100 MHz
                if (give && __hip_atomic_load(keepalive_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == epoch) {
                    __hip_atomic_store(keepalive_host, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (used up: the host writes it again while it still holds)
                    tMoved = tNow; give = false;
                }
                shGive = give ? 1u : 0u;
            }
            __syncthreads();
            const bool gaveUp = shGive != 0u;
            __syncthreads();                                            // (shGive is rewritten in the next round)
            if (gaveUp) last = true;
            const unsigned word = (unsigned) complete | (last ? RESOLVE_LAST : 0u) | (gaveUp ? RESOLVE_GAVE_UP : 0u);
            if (word != published) {
                if (tid == 0) atomicExch(leaderWord, word);
#ifdef JTX_DBG_PROG
                if (tid == 0) { unsigned *lg = leaderWord + 16; const unsigned i = lg[0]; if (i < 14) { lg[1 + 3 * i] = (unsigned) (__builtin_amdgcn_s_memrealtime() / 100); lg[2 + 3 * i] = dealt / (unsigned) p.num_subblocks; lg[3 + 3 * i] = allGone ? 9999u : (oldest - p.sample_begin) / p.strata_per_group; lg[0] = i + 1; } }
#endif
                published = word;
            }
            if (last) break;
            __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);      // ~15 us a round: a few hundred rounds per pass of C2
        }
        return;
    }
    // ---- the workers: passes [done, complete) into the film of this workgroup's pixels ----
    const int nworkers = nwg > 1 ? nwg - 1 : 1, me = nwg > 1 ? wg - 1 : 0;
    int done = 0;                                                       // passes this workgroup has added for all its pixels
    while (true) {
        // (an atomic: the word as it is in memory now -- on a cache line of its own, never the chunk counter's)
        if (tid == 0) shWord = atomicOr(leaderWord, 0u);
        __syncthreads();
        const unsigned word = shWord;
        __syncthreads();
        const int complete = (int) (word & ~(RESOLVE_LAST | RESOLVE_GAVE_UP));
        if (word & RESOLVE_GAVE_UP) {                                   // (the host reads it in every workgroup's word: bit 15 of the count)
            if (tid == 0) __hip_atomic_store(progress_host + wg, (epoch << 16) | 0x8000u | (unsigned) done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        if (complete > done) {
            const int sA = p.sample_begin + done * p.strata_per_group;
            int sB = p.sample_begin + complete * p.strata_per_group; if (sB > p.sample_end) sB = p.sample_end;
            // two pixels per round and thread, the loads of both in flight together; the sums in sample order (image.hpp:82-86)
            for (int pslot = me * RESOLVE_BLOCK + tid; pslot < p.rad_stride; pslot += 2 * nworkers * RESOLVE_BLOCK) {
                size_t pixA = 0, pixB = 0;
                const int pslotB = pslot + nworkers * RESOLVE_BLOCK;
                const bool inA = resolveSlot(p, pslot, pixA), inB = resolveSlot(p, pslotB, pixB);
                f3 accA = mk3(0.0f), accB = mk3(0.0f);
                if (sA > 0) {
                    if (inA) accA = mk3(p.acc[3 * pixA], p.acc[3 * pixA + 1], p.acc[3 * pixA + 2]);
                    if (inB) accB = mk3(p.acc[3 * pixB], p.acc[3 * pixB + 1], p.acc[3 * pixB + 2]);
                }
                // (slots of the padding are loaded too: the buffer has them; slot B of the last round may lie behind it: the first slot then)
                const float4 *rA = p.rad + ((size_t) (sA - p.sample_begin) * p.rad_stride + pslot);
                const float4 *rB = p.rad + ((size_t) (sA - p.sample_begin) * p.rad_stride + (pslotB < p.rad_stride ? pslotB : pslot));
                int s = sA;
                for (; s + 2 <= sB; s += 2, rA += (size_t) 2 * p.rad_stride, rB += (size_t) 2 * p.rad_stride) {
                    u32x4 a0, a1, b0, b1;
                    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                                 "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1)
                                 : "v"(rA), "v"(rA + p.rad_stride), "v"(rB), "v"(rB + p.rad_stride) : "memory");
                    accA = accA + mk3(__uint_as_float(a0.x), __uint_as_float(a0.y), __uint_as_float(a0.z));
                    accA = accA + mk3(__uint_as_float(a1.x), __uint_as_float(a1.y), __uint_as_float(a1.z));
                    accB = accB + mk3(__uint_as_float(b0.x), __uint_as_float(b0.y), __uint_as_float(b0.z));
                    accB = accB + mk3(__uint_as_float(b1.x), __uint_as_float(b1.y), __uint_as_float(b1.z));
                }
                for (; s < sB; ++s, rA += p.rad_stride, rB += p.rad_stride) {
                    u32x4 a0, b0;
                    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(a0), "=&v"(b0) : "v"(rA), "v"(rB) : "memory");
                    accA = accA + mk3(__uint_as_float(a0.x), __uint_as_float(a0.y), __uint_as_float(a0.z));
                    accB = accB + mk3(__uint_as_float(b0.x), __uint_as_float(b0.y), __uint_as_float(b0.z));
                }
                if (inA) resolveStore(p, pixA, accA, sB);
                if (inB) resolveStore(p, pixB, accB, sB);
            }
            done = complete;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                            // every pixel of the workgroup: then the host may hear of it
            if (tid == 0) __hip_atomic_store(progress_host + wg, (epoch << 16) | (unsigned) done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (word & RESOLVE_LAST) break;
        else __builtin_amdgcn_s_sleep(127);
    }
}
} // namespace jtx

hipError_t jtx_launch_resolve_progressive(const RenderParams &p, int num_owned_tiles, int num_path_waves, int num_workgroups, unsigned *started_host,
                                          unsigned *progress_host, unsigned *keepalive_host, unsigned epoch, hipStream_t stream, unsigned patience_ms) {
    if (num_owned_tiles <= 0 || num_workgroups <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_resolve_progressive, dim3((unsigned) num_workgroups), dim3(RESOLVE_BLOCK), 0, stream, p, num_path_waves, started_host, progress_host,
                       keepalive_host, epoch, patience_ms ? patience_ms : RESOLVE_PATIENCE_MS);
    return hipGetLastError();
}

hipError_t jtx_launch_resolve_samples(const RenderParams &p, int num_owned_tiles, hipStream_t stream) {
    if (num_owned_tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_resolve_samples, dim3((unsigned) (num_owned_tiles * 1024 / BLOCK)), dim3(BLOCK), 0, stream, p);
    return hipGetLastError();
}

static inline unsigned blocksFor(int n) { return (unsigned) ((n + BLOCK - 1) / BLOCK); }

// The BVH source the TIMED launch of this scene walks (jtx_launch_render_paths' choice): what traversal = 1 of the per-ray entry points runs
int jtx_production_source(const DevScene &sc) {
    const bool lds = sc.lds_threaded != 0;
    static const int leafWalk = [] { const char *e = getenv("JTX_LEAF_WALK"); return e ? atoi(e) : 1; }();
    if (lds) return (sc.lw_leaves > 0 && leafWalk) ? SRC_LEAF : SRC_LDS;
    return sc.wide != nullptr ? SRC_WIDE : SRC_GLOBAL;
}

static size_t batchShmem(const DevScene &sc, int src, int bs) {
    if (src == SRC_WIDE) return (size_t) sc.wide_depth * bs * sizeof(uint2);
    if (src == SRC_LDS) return ldsBytes(sc, true);
    if (src == SRC_LEAF) return ldsBytes(sc, true) + (size_t) 2 * ((sc.lw_leaves + 3) & ~3) * sizeof(float4) + 128 * sizeof(unsigned);
    return 0;
}

// src: SRC_GLOBAL / SRC_LDS / SRC_LEAF / SRC_WIDE (the caller has checked that the scene carries that structure).  The 8-ary instance
// runs single-wave workgroups like the timed kernel (stack stride 64), the LDS-staged ones its 256-lane workgroups.
hipError_t jtx_launch_closest_batch(const DevScene &sc, int src, int n, const float *o, const float *d, float tmin, float tmax,
                                    int *hit, float *t, int *prim, float *b1, float *b2, float *point, float *normal,
                                    float *uv, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int bs = src == SRC_WIDE ? 64 : BLOCK;
    const dim3 grid((unsigned) ((n + bs - 1) / bs)), block(bs);
    const size_t shmem = batchShmem(sc, src, bs);
#define LAUNCH_CB(S, B) hipLaunchKernelGGL((k_closest_batch<S, B>), grid, block, shmem, stream, sc, n, o, d, tmin, tmax, hit, t, prim, b1, b2, point, normal, uv)
    if (src == SRC_WIDE) LAUNCH_CB(SRC_WIDE, 64); else if (src == SRC_LEAF) LAUNCH_CB(SRC_LEAF, BLOCK); else if (src == SRC_LDS) LAUNCH_CB(SRC_LDS, BLOCK); else LAUNCH_CB(SRC_GLOBAL, BLOCK);
#undef LAUNCH_CB
    return hipGetLastError();
}
hipError_t jtx_launch_any_batch(const DevScene &sc, int src, int n, const float *o, const float *d, const float *tmin,
                                const float *tmax, int *hit, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int bs = src == SRC_WIDE ? 64 : BLOCK;
    const dim3 grid((unsigned) ((n + bs - 1) / bs)), block(bs);
    const size_t shmem = batchShmem(sc, src, bs);
#define LAUNCH_AB(S, B) hipLaunchKernelGGL((k_any_batch<S, B>), grid, block, shmem, stream, sc, n, o, d, tmin, tmax, hit)
    if (src == SRC_WIDE) LAUNCH_AB(SRC_WIDE, 64); else if (src == SRC_LEAF) LAUNCH_AB(SRC_LEAF, BLOCK); else if (src == SRC_LDS) LAUNCH_AB(SRC_LDS, BLOCK); else LAUNCH_AB(SRC_GLOBAL, BLOCK);
#undef LAUNCH_AB
    return hipGetLastError();
}
hipError_t jtx_launch_bxdf_batch(const DevScene &sc, int mode, int material, int n, const float *normal, const float *uv,
                                 const float *wo, const float *wi_in, const float *uc, const float *u2, int *ok, float *f,
                                 float *wi_out, float *pdf, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_bxdf_batch, dim3(blocksFor(n)), dim3(BLOCK), 0, stream, sc, mode, material, n, normal, uv, wo, wi_in,
                       uc, u2, ok, f, wi_out, pdf);
    return hipGetLastError();
}
hipError_t jtx_launch_camera_rays(const DCam &cam, int n, const int *row, const int *col, const int *sample, float *o,
                                  float *d, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_camera_rays, dim3(blocksFor(n)), dim3(BLOCK), 0, stream, cam, n, row, col, sample, o, d);
    return hipGetLastError();
}
hipError_t jtx_launch_radiance_samples(const DevScene &sc, const DCam &cam, int maxDepth, int n, const int *row,
                                       const int *col, const int *sample, float *rgb, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_radiance_samples, dim3(blocksFor(n)), dim3(BLOCK), ldsBytes(sc, false), stream, sc, cam, maxDepth, n,
                       row, col, sample, rgb);
    return hipGetLastError();
}
hipError_t jtx_launch_rng_stream(uint32_t x, uint32_t y, uint32_t n, int count, uint32_t *out_u32, float *out_f32,
                                 hipStream_t stream) {
    hipLaunchKernelGGL(k_rng_stream, dim3(1), dim3(64), 0, stream, x, y, n, count, out_u32, out_f32);
    return hipGetLastError();
}
hipError_t jtx_launch_sincos(const float *x, int n, float *s, float *c, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_sincos, dim3(blocksFor(n)), dim3(BLOCK), 0, stream, x, n, s, c);
    return hipGetLastError();
}
