// jtx_wavefront.hip -- the wavefront integrator: integrateMIS (integrator.cpp:171-216) split into
// stages that exchange SoA path / ray / hit buffers resident in HBM.
//
//   k_wf_generate : Camera::getRay for a batch of strata           -> ray[slot], path state[slot]
//   k_wf_trace<0> : Scene::closestHit over all live slots           -> hit[slot]
//   k_wf_shade    : hit -> sky / depth stop / light sample / BSDF   -> shadow ray + pending radiance,
//                                                                      next ray, path state
//   k_wf_trace<1> : Scene::anyHit over pending shadow rays          -> radiance[slot] += pending
//   k_wf_resolve  : per pixel, strata in order: clamp + AccumulationBuffer::updatePixel + setPixel
//
// A slot is one path (pixel, stratum) of the batch; slot index = stratum_local * pixels + pixel, the
// pixel order being the 8x8-block order of the tile mapping, so neighbouring lanes hold neighbouring
// pixels.  Slots never move: every stage reads and writes slot-indexed SoA arrays (coalesced), and
// a stage finds its work by scanning a per-slot flag word.  There is NO global compaction queue and
// no atomic at all: a persistent wave owns an interleaved set of 64-slot groups, reads 64 flags at a
// time (one coalesced load + one ballot) and hands live slots to whichever of its lanes are idle
// (wave-ballot compaction).  In the trace kernels a lane that finishes its ray is refilled at once,
// so the 64 lanes stay busy although ray lengths differ by 10x; that is what the register-resident
// pixel-persistent kernel cannot do.
//
// Every path's arithmetic is self-contained and identical to the oracle's, so results do not depend
// on which wave or lane processes a slot: output is bit-identical to the pixel-persistent kernel.
#include "jtx_scene_dev.hpp"
#include "jtx_launch.hpp"

namespace jtx {

constexpr int WBLOCK = 256;
#ifndef JTX_REFILL_VOTE
#define JTX_REFILL_VOTE 16       // lanes without a prefetched ray that trigger a prefetch in the trace kernels
#endif
#ifndef JTX_RETIRE_VOTE
#define JTX_RETIRE_VOTE 8        // finished lanes that end the interior phase (retire + switch to the prefetched ray)
#endif

// ---- per-wave slot fetcher ----------------------------------------------------------------------
// Work distribution: the slot range is cut into 64-slot groups (= one 8x8 pixel block of one stratum)
// and wave w of the grid owns groups w, w + nwaves, w + 2*nwaves, ...  No atomics: a shared head
// word cost ~11 ns per grab, i.e. ~200 us per launch for 8192 waves, 25x the useful work of a late
// round.  Interleaving spreads sky / interior regions over all waves, which balances them
// statistically.
struct WaveFetch {
    // wave-uniform bookkeeping (SGPRs)
    int gnext;                         // group index whose flags will be requested next
    int gstride;                       // waves in the grid
    int gbase;                         // first slot of the current 64-slot group
    int nbase;                         // first slot of the group whose flags are in flight (nflag)
    unsigned long long gmask;          // live, not yet handed-out slots of the current group
    bool drained;
    int nflag;                         // per lane: prefetched flag word of slot nbase + lane
    JD void request(const int *flags, int num_slots) {     // issue the (coalesced) flag load of the next group
        nbase = gnext * 64;
        gnext += gstride;
        const int s = nbase + (threadIdx.x & 63);
        nflag = (s < num_slots) ? flags[s] : 0;
    }
    JD void init(const int *flags, int num_slots) {
        gnext = blockIdx.x * (WBLOCK / 64) + (threadIdx.x >> 6);
        gstride = gridDim.x * (WBLOCK / 64);
        gbase = 0; gmask = 0ull; drained = false;
        request(flags, num_slots);
    }
};

// Hands live slots (flags[slot] & want) to the lanes with need == true.  Returns the slot or -1.
// `scratch` is 64 ints of LDS private to the wave.  The flag words of the following group are
// requested as soon as a group is opened, so the scan never waits on HBM in steady state.
JD int waveFetch(WaveFetch &w, bool need, const int *flags, int want, int num_slots, int *scratch,
                 int mask2 = 0, int value2 = 0, int value3 = -1) {
    const int lane = threadIdx.x & 63;
    int slot = -1;
    unsigned long long needMask = __ballot(need);
    while (needMask != 0ull) {
        if (w.gmask == 0ull) {
            if (w.drained) break;
            if (w.nbase >= num_slots) { w.drained = true; break; }
            w.gbase = w.nbase;
            // live test; the shade launches additionally select by the hit-material tag in bits 8-11
            w.gmask = __ballot((w.nflag & want) != 0 && (mask2 == 0 || (w.nflag & mask2) == value2 || (w.nflag & mask2) == value3));
            w.request(flags, num_slots);
            continue;
        }
        // k-th needy lane takes the k-th live slot of the group (wave-ballot compaction)
        const int nNeed = __popcll(needMask), nLive = __popcll(w.gmask);
        const int n = nNeed < nLive ? nNeed : nLive;
        const unsigned long long below = (1ull << lane) - 1ull;
        if ((w.gmask >> lane) & 1ull) {
            const int r = __popcll(w.gmask & below);
            if (r < n) scratch[r] = lane;
        }
        __builtin_amdgcn_wave_barrier();
        const bool iNeed = (needMask >> lane) & 1ull;
        const int myRank = __popcll(needMask & below);
        if (iNeed && myRank < n) slot = w.gbase + scratch[myRank];
        __builtin_amdgcn_wave_barrier();
        // drop the n lowest set bits of both masks (wave-uniform)
        unsigned long long gm = w.gmask, nm = needMask;
        for (int i = 0; i < n; ++i) { gm &= gm - 1ull; nm &= nm - 1ull; }
        w.gmask = gm; needMask = nm;
    }
    return slot;
}

// ---- slot <-> pixel mapping (same 32x32 tile / 8x8 wave-block order as k_render_pixels) ----------
JD bool slotPixel(const WfParams &p, int pix, int &row, int &col) {
    const int owned = pix >> 10, sub = (pix >> 6) & 15, lane = pix & 63;
    const int tile = p.tile_rank + owned * p.tile_world;
    const int trow = tile / p.tiles_x, tcol = tile - trow * p.tiles_x;
    row = trow * 32 + (sub >> 2) * 8 + (lane >> 3);
    col = tcol * 32 + (sub & 3) * 8 + (lane & 7);
    return row < p.height && col < p.width;
}

// ---- generate ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WBLOCK) k_wf_generate(WfParams p, int s0, int nstrata) {
    const int slot = blockIdx.x * WBLOCK + threadIdx.x;
    if (slot >= p.num_slots) return;
    const int sl = slot / p.pixels, pix = slot - sl * p.pixels;
    int row, col;
    int flag = 0;
    if (sl < nstrata && slotPixel(p, pix, row, col)) {
        Rng rng; rng.seed(row, col, (uint32_t) (s0 + sl) + 1u);                  // camera.cpp:101
        f3 o, d; cameraRay(p.cam, col, row, s0 + sl, rng, o, d);
        p.b.rox[slot] = o.x; p.b.roy[slot] = o.y; p.b.roz[slot] = o.z;
        p.b.rdx[slot] = d.x; p.b.rdy[slot] = d.y; p.b.rdz[slot] = d.z;
        p.b.betax[slot] = 1.0f; p.b.betay[slot] = 1.0f; p.b.betaz[slot] = 1.0f;
        p.b.radx[slot] = 0.0f; p.b.rady[slot] = 0.0f; p.b.radz[slot] = 0.0f;
        p.b.rng[slot] = rng.state; p.b.depth[slot] = 0;
        flag = WF_LIVE;
    }
    p.b.flags[slot] = flag;
    p.b.sflags[slot] = 0;
    if (p.counters) {
        const unsigned long long m = __ballot(flag != 0);
        if ((threadIdx.x & 63) == __ffsll((long long) m) - 1) atomicAdd(&p.counters[0], (unsigned long long) __popcll(m));
    }
}

// ---- trace ---------------------------------------------------------------------------------------
// Persistent.  Per lane: an ACTIVE ray (slot >= 0) that is WALKing (cur >= 0, leafN == 0), PARKED on
// a leaf (leafN > 0) or FINISHED (cur < 0), plus one PREFETCHED ray (pslot >= 0) whose o/d loads were
// issued an outer iteration earlier: a lane that finishes switches to its prefetched ray without
// waiting on HBM, and the wave never blocks on a refill.
// WIDE: HBM-resident scene, uncounted: the rays walk the 8-ary quantised nodes (traverseWide's steps, per-lane stack
// in LDS behind the scan scratch); rays those nodes cannot take are traced on the binary records when activated.
#ifndef JTX_WF_WIDE_OCC
#define JTX_WF_WIDE_OCC 1
#endif
template <int ANY, bool COUNT, bool LDS_SCENE, bool WIDE>
__global__ void __launch_bounds__(WBLOCK, WIDE ? JTX_WF_WIDE_OCC : 1) k_wf_trace(WfParams p) {
    static_assert(!WIDE || (!COUNT && !LDS_SCENE), "wide nodes: uncounted kernels of HBM-resident scenes only");
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const DevScene &sc = p.scene;
    // LDS: [wave scratch: WBLOCK ints][8 threaded node orderings][tris]  -- the traversal is stackless
    int *scratchAll = smem;
    float4 *lds_tnodes = (float4 *) (scratchAll + WBLOCK);
    float4 *lds_tris = lds_tnodes + 2 * 8 * sc.num_nodes;
    if (LDS_SCENE) {
        const int nn = 2 * 8 * sc.num_nodes, nt = 3 * sc.num_prims;
        for (int i = threadIdx.x; i < nn; i += WBLOCK) lds_tnodes[i] = sc.tnodes[i];
        for (int i = threadIdx.x; i < nt; i += WBLOCK) lds_tris[i] = sc.tris[i];
        __syncthreads();
    }
    const float4 *tnodes = LDS_SCENE ? lds_tnodes : sc.tnodes;
    const float4 *tris = LDS_SCENE ? lds_tris : sc.tris;
    int *scratch = scratchAll + (threadIdx.x & ~63);
    uint2 *stk = (uint2 *) (scratchAll + WBLOCK) + threadIdx.x;       // WIDE: this lane's stack column (stride WBLOCK)
    const int *flags = ANY ? p.b.sflags : p.b.flags;
    const int want = ANY ? WF_SH_PENDING : WF_LIVE;
    const float tmin = ANY ? 0.0f : 0.001f;
    WideRay wr; WideState ws; ws.start(); ws.done = true;
    wideRaySetup(wr, mk3(0.0f), mk3(1.0f), mk3(1.0f), 0, tmin, 0.0f);

    Counters9 cnt = {};
    if (sc.num_nodes == 0) {
        // no geometry: every closest ray misses, every shadow ray is unoccluded
        for (int s = blockIdx.x * WBLOCK + threadIdx.x; s < p.num_slots; s += gridDim.x * WBLOCK) {
            if (!(flags[s] & want)) continue;
            if (ANY) { p.b.sflags[s] = WF_SH_UNOCCLUDED; if (COUNT) cnt.n_any++; }   // no geometry: nothing occludes
            else { p.b.hit[s] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1)); p.b.flags[s] = WF_LIVE; if (COUNT) cnt.n_closest++; }
        }
        if (COUNT) { const unsigned long long a = ANY ? cnt.n_any : cnt.n_closest; if (a) atomicAdd(&p.counters[ANY ? 2 : 1], a); }
        return;
    }

    WaveFetch wf; wf.init(flags, p.num_slots);
    int slot = -1, cur = -1, leafOff = 0, leafN = 0, negmask = 0;      // leafN: count | last-record flag (threaded records)
    bool hitAny = false;
    f3 o = mk3(0.0f), d = mk3(1.0f), inv = mk3(1.0f);
    float tmax = 0.0f;
    HitRec rec; rec.t = 0.0f; rec.prim = -1; rec.b1 = rec.b2 = 0.0f;
    int pslot = -1;
    float pox = 0.0f, poy = 0.0f, poz = 0.0f, pdx = 1.0f, pdy = 1.0f, pdz = 1.0f, ptmax = 0.0f;
    int pnf = 0, nf = 0;                            // shadow rays: the beta-non-finite bits travelling with the slot

    while (true) {
        // ---- A. switch finished / empty lanes to their prefetched ray ----
        if (slot < 0 && pslot >= 0) {
            slot = pslot; pslot = -1; nf = pnf;
            o = mk3(pox, poy, poz); d = mk3(pdx, pdy, pdz);
            tmax = ANY ? ptmax : __builtin_inff();
            inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            negmask = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
            const bool regular = finiteNonZero(inv.x) && finiteNonZero(inv.y) && finiteNonZero(inv.z) &&
                                 fabsf(o.x) < __builtin_inff() && fabsf(o.y) < __builtin_inff() && fabsf(o.z) < __builtin_inff() &&
                                 tmax == tmax;
            cur = negmask * sc.num_nodes; leafN = 0; hitAny = false; rec.prim = -1; rec.t = 0.0f; rec.b1 = rec.b2 = 0.0f;
            if (WIDE) { wideRaySetup(wr, o, d, inv, negmask, tmin, tmax); ws.start(); }
            if (!regular || (WIDE && !wideRayOk(o, inv, tmin, tmax))) {
                // axis-parallel / non-finite rays: the exact slab test, traced to the end right here
                // (rare; keeps the main loop on the min/max form only)
                GlobalSrc src; src.tnodes = tnodes; src.tris = tris;
                hitAny = traverseThreaded<ANY != 0, COUNT, false>(src, sc.num_nodes, o, d, inv, negmask, tmin, tmax, rec, cnt);
                cur = -1;
                if (WIDE) { ws.done = true; ws.hitAnything = hitAny; }
            } else if (COUNT) { if (ANY) cnt.n_any++; else cnt.n_closest++; }
        }
        // ---- B. prefetch the next ray of lanes that hold none ----
        const unsigned long long active = __ballot(slot >= 0);
        if (!wf.drained) {
            const unsigned long long needP = __ballot(pslot < 0);
            if (__popcll(needP) >= JTX_REFILL_VOTE || active == 0ull) {
                const int s = waveFetch(wf, pslot < 0, flags, want, p.num_slots, scratch);
                if (s >= 0) {
                    pslot = s;
                    if (ANY) { pox = p.b.sox[s]; poy = p.b.soy[s]; poz = p.b.soz[s]; pdx = p.b.sdx[s]; pdy = p.b.sdy[s]; pdz = p.b.sdz[s]; ptmax = p.b.stmax[s];
                               pnf = p.b.sflags[s] & (WF_SH_NF_MASK | WF_SH_TYPE_MASK); }
                    else     { pox = p.b.rox[s]; poy = p.b.roy[s]; poz = p.b.roz[s]; pdx = p.b.rdx[s]; pdy = p.b.rdy[s]; pdz = p.b.rdz[s]; }
                }
            }
        }
        if (active == 0ull) {
            if (__ballot(pslot >= 0) == 0ull) break;         // wave-uniform: nothing left anywhere
            continue;                                        // go activate what was just fetched
        }

        if constexpr (WIDE) {
            // ---- C/D (wide nodes): node steps until enough lanes are parked on a leaf or finished; then the leaves ----
            while (true) {
                if (slot >= 0 && ws.walking()) wideNodeStep<ANY == 0>(sc.wide, stk, WBLOCK, wr, ws);
                const unsigned long long walking = __ballot(slot >= 0 && ws.walking());
                const unsigned long long parked = __ballot(slot >= 0 && ws.pendLeaf >= 0);
                const unsigned long long finished = __ballot(slot >= 0 && ws.done && ws.pendLeaf < 0);
                if (walking == 0ull || __popcll(parked) >= JTX_WIDE_LEAF_VOTE || __popcll(finished) >= JTX_RETIRE_VOTE) break;
                if (parked != 0ull && __popcll(walking) <= JTX_WIDE_FEW_WALKERS) break;
            }
            if (slot >= 0 && ws.pendLeaf >= 0) {
                GlobalSrc src; src.tnodes = tnodes; src.tris = tris;
                wideLeafStep(sc.wide, src, ANY != 0, wr, ws, rec);
            }
            if (slot >= 0 && ws.done && ws.pendLeaf < 0) { cur = -1; leafN = 0; hitAny = ws.hitAnything; }
            else if (slot >= 0) cur = 0;                                  // still under way
        } else {
        // ---- C. interior phase: one node per walking lane per iteration ----
            while (true) {
#pragma unroll
                for (int rep = 0; rep < JTX_STEPS_PER_VOTE; ++rep) {
                    if (slot >= 0 && leafN == 0 && cur >= 0) {
                        const float4 na = tnodes[2 * cur], nb = tnodes[2 * cur + 1];
                        if (COUNT) { if (ANY) cnt.n_nodes_any++; else cnt.n_nodes_closest++; }
                        const bool boxHit = slabRegular(na, nb, o, inv, tmin, tmax);
                        const int w = __float_as_int(nb.w), z = __float_as_int(nb.z);
                        if (boxHit && w != 0) { leafN = w; leafOff = z; }                    // park on the leaf
                        else cur = (boxHit || w != 0) ? (w < 0 ? -1 : cur + 1) : z;          // next record / skip link
                    }
                }
                const unsigned long long walking = __ballot(slot >= 0 && leafN == 0 && cur >= 0);
                const unsigned long long parked = __ballot(leafN != 0);
                const unsigned long long finished = __ballot(slot >= 0 && leafN == 0 && cur < 0);
                if (walking == 0ull || __popcll(parked) >= JTX_LEAF_VOTE || __popcll(finished) >= JTX_RETIRE_VOTE) break;
            }

            // ---- D. leaf phase ----
            if (leafN != 0) {
                GlobalSrc src; src.tris = tris;
                const int n = leafN & 0xffff;
                for (int i = 0; i < n; ++i) {
                    const int prim = leafOff + i;
                    if (COUNT) { if (ANY) cnt.n_tri_any++; else cnt.n_tri_closest++; }
                    float b1, b2, root;
                    if (!triTest(src, prim, o, d, tmin, tmax, b1, b2, root)) continue;
                    hitAny = true;
                    if (ANY) break;
                    tmax = root;
                    rec.t = root; rec.prim = prim; rec.b1 = b1; rec.b2 = b2;
                    if (COUNT) cnt.n_accept++;
                }
                cur = ((ANY && hitAny) || leafN < 0) ? -1 : cur + 1;
                leafN = 0;
            }
        }

        // ---- E. retire finished rays: one store, nothing to wait for ----
        if (slot >= 0 && cur < 0 && leafN == 0) {
            if (ANY) { p.b.sflags[slot] = hitAny ? (nf & WF_SH_NF_MASK) : WF_SH_UNOCCLUDED;    // shade / resolve add the pending radiance (or poison)
                       if (COUNT && !hitAny) countClass(cnt.n_eval_t, (nf & WF_SH_TYPE_MASK) >> WF_SH_TYPE_SHIFT); }   // sampleLights reached evalBxdf (integrator.cpp:151-166)
            else {
                p.b.hit[slot] = make_float4(rec.t, rec.b1, rec.b2, __int_as_float(hitAny ? rec.prim : -1));
                if (p.sort_shade) {                                        // shading sorted by the hit material's type
                    int tag = 0;
                    if (hitAny) tag = (__float_as_int(tris[3 * rec.prim + 2].y) + 1) << WF_TYPE_SHIFT;
                    p.b.flags[slot] = WF_LIVE | tag;
                }
            }
            slot = -1;
        }
    }
    if (COUNT) {
        const unsigned v[4] = {ANY ? cnt.n_any : cnt.n_closest, ANY ? cnt.n_nodes_any : cnt.n_nodes_closest,
                               ANY ? cnt.n_tri_any : cnt.n_tri_closest, cnt.n_accept};
        const int idx[4] = {ANY ? 2 : 1, ANY ? 6 : 3, ANY ? 7 : 4, 5};
        for (int i = 0; i < 4; ++i) {
            unsigned long long s = v[i];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if ((threadIdx.x & 63) == 0 && s) atomicAdd(&p.counters[idx[i]], s);
        }
        if (ANY) for (int i = 0; i < 7; ++i) {
            unsigned long long s = i == 0 ? cnt.n_eval_t[0] : i == 1 ? cnt.n_eval_t[1] : i == 2 ? cnt.n_eval_t[2] : i == 3 ? cnt.n_eval_t[3] : i == 4 ? cnt.n_eval_t[4] : i == 5 ? cnt.n_eval_t[5] : cnt.n_eval_t[6];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if ((threadIdx.x & 63) == 0 && s) atomicAdd(&p.counters[CNT_EVAL_T + i], s);
        }
    }
}

// ---- shade ---------------------------------------------------------------------------------------
// Persistent; lanes are filled with live slots by the wave fetcher, so the shading code runs on full
// waves even when most paths of the batch have already ended.
// typeCode: -1 = every live slot; else low 4 bits = 1 + Material::type to take, bit 4 = also take the misses.
template <bool COUNT, int MASK>
__global__ void __launch_bounds__(WBLOCK) k_wf_shade(WfParams p, int typeCode) {
    __shared__ int scratchAll[WBLOCK];
    const DevScene &sc = p.scene;
    int *scratch = scratchAll + (threadIdx.x & ~63);
    WaveFetch wf; wf.init(p.b.flags, p.num_slots);
    unsigned nshade = 0, nshadeT[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    while (true) {
        const int slot = typeCode < 0 ? waveFetch(wf, true, p.b.flags, WF_LIVE, p.num_slots, scratch)
                                      : waveFetch(wf, true, p.b.flags, WF_LIVE, p.num_slots, scratch, WF_TYPE_MASK,
                                                  (typeCode & 15) << WF_TYPE_SHIFT, (typeCode & 16) ? 0 : -1);
        if (__ballot(slot >= 0) == 0ull) break;
        if (slot < 0) continue;
        const float4 hv = p.b.hit[slot];
        const int prim = __float_as_int(hv.w);
        f3 rad = mk3(p.b.radx[slot], p.b.rady[slot], p.b.radz[slot]);
        f3 beta = mk3(p.b.betax[slot], p.b.betay[slot], p.b.betaz[slot]);
        int flag = 0;                                                     // path ends unless set below
        int sflag = 0;
        bool radDirty = false;
        const int sprev = p.b.sflags[slot];
        if (sprev & WF_SH_UNOCCLUDED) {                                   // the previous vertex' light sample got through
            rad = rad + mk3(p.b.pendx[slot], p.b.pendy[slot], p.b.pendz[slot]);   // integrator.cpp:194-196
            radDirty = true;
        } else if (sprev & WF_SH_NF_MASK) {                               // occluded, but beta was inf/NaN: beta * {} is NaN
            rad = poisonNonFinite(rad, (sprev >> WF_SH_NF_SHIFT) & 7);
            radDirty = true;
        }
        if (prim < 0) {                                                   // integrator.cpp:183-187
            rad = rad + beta * a3(sc.sky);
            radDirty = true;
        } else {
            int depth = p.b.depth[slot];
            if (depth++ != p.max_depth) {                                 // integrator.cpp:191
                const f3 o = mk3(p.b.rox[slot], p.b.roy[slot], p.b.roz[slot]);
                const f3 d = mk3(p.b.rdx[slot], p.b.rdy[slot], p.b.rdz[slot]);
                HitRec h; h.t = hv.x; h.b1 = hv.y; h.b2 = hv.z; h.prim = prim;
                const Surface sf = makeSurface(sc.shade, h, o, d);
                const DMaterial &mat = sc.materials[sf.material];
                ShadeCtx ctx; ctx.materials = sc.materials; ctx.textures = sc.textures; ctx.texels = sc.texels;
                Rng rng; rng.state = p.b.rng[slot];
                const f3 wo = -d;
                if (sc.num_lights > 0) {                                  // sampleLights integrator.cpp:134-169
                    const uint32_t idx = rng.sampleRange(sc.num_lights - 1);
                    const DLight &light = sc.lights[idx];
                    (void) rng.f(); (void) rng.f();
                    LightSample ls;
                    if (lightSample(light, sf.point, ls)) {
                        const f3 so = sf.point + sf.normal * RAY_EPSILON;
                        const float lDist = len(sf.point - ls.p);
                        // the contribution an unoccluded shadow ray will add (beta of THIS vertex)
                        f3 f; float pb;
                        evalPdfBxdf<MASK>(ctx, mat, sf.normal, sf.uv, wo, ls.wi, f, pb);
                        f = f * absdot(ls.wi, sf.normal);
                        const float pl = 1.0f / (float) sc.num_lights * ls.pdf;
                        const float misWeight = powerHeuristic(1.0f, pl, 1.0f, pb);
                        const f3 pend = beta * (misWeight * f * ls.radiance / pl);
                        p.b.sox[slot] = so.x; p.b.soy[slot] = so.y; p.b.soz[slot] = so.z;
                        p.b.sdx[slot] = ls.wi.x; p.b.sdy[slot] = ls.wi.y; p.b.sdz[slot] = ls.wi.z;
                        p.b.stmax[slot] = lDist - RAY_EPSILON;
                        p.b.pendx[slot] = pend.x; p.b.pendy[slot] = pend.y; p.b.pendz[slot] = pend.z;
                        sflag = WF_SH_PENDING | (nonFiniteMask(beta) << WF_SH_NF_SHIFT) | (bxdfClass(mat) << WF_SH_TYPE_SHIFT);
                    }
                }
                const float u = rng.f();
                f2 u2; u2.x = rng.f(); u2.y = rng.f();
                BSample bs;
                if (COUNT) { nshade++; countClass(nshadeT, bxdfClass(mat)); }
                if (sampleBxdf<MASK>(ctx, mat, sf.normal, sf.uv, wo, u, u2, bs)) {
                    if (bs.pdf > 0.0f) beta = beta * (bs.f * absdot(bs.wi, sf.normal) / bs.pdf);
                    const f3 no = sf.point + bs.wi * RAY_EPSILON;         // integrator.cpp:212
                    p.b.rox[slot] = no.x; p.b.roy[slot] = no.y; p.b.roz[slot] = no.z;
                    p.b.rdx[slot] = bs.wi.x; p.b.rdy[slot] = bs.wi.y; p.b.rdz[slot] = bs.wi.z;
                    p.b.betax[slot] = beta.x; p.b.betay[slot] = beta.y; p.b.betaz[slot] = beta.z;
                    p.b.rng[slot] = rng.state; p.b.depth[slot] = depth;
                    flag |= WF_LIVE;
                }
            }
        }
        if (radDirty) { p.b.radx[slot] = rad.x; p.b.rady[slot] = rad.y; p.b.radz[slot] = rad.z; }
        p.b.flags[slot] = flag;
        p.b.sflags[slot] = sflag;
    }
    if (COUNT) {
        unsigned long long s = nshade;
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&p.counters[8], s);
        for (int i = 0; i < 7; ++i) {
            unsigned long long st = i == 0 ? nshadeT[0] : i == 1 ? nshadeT[1] : i == 2 ? nshadeT[2] : i == 3 ? nshadeT[3] : i == 4 ? nshadeT[4] : i == 5 ? nshadeT[5] : nshadeT[6];
            for (int off = 32; off > 0; off >>= 1) st += __shfl_down(st, off, 64);
            if ((threadIdx.x & 63) == 0 && st) atomicAdd(&p.counters[CNT_SHADE_T + i], st);
        }
    }
}

// ---- resolve -------------------------------------------------------------------------------------
JD unsigned char wfToByte(float v) {                                       // image.hpp:9-16,47-52
    const float g = v > 0.0f ? sqrtf(v) : 0.0f;
    const float c = clampf(g, 0.0f, 0.999f);
    return (unsigned char) (int) (255.999f * c);
}

__global__ void __launch_bounds__(WBLOCK) k_wf_resolve(WfParams p, int s0, int nstrata, int write_img) {
    const int pix = blockIdx.x * WBLOCK + threadIdx.x;
    if (pix >= p.pixels) return;
    int row, col;
    if (!slotPixel(p, pix, row, col)) return;
    const size_t px = (size_t) row * p.width + col;
    f3 acc = mk3(0.0f);
    if (s0 > 0) acc = mk3(p.acc[3 * px], p.acc[3 * px + 1], p.acc[3 * px + 2]);
    for (int sl = 0; sl < nstrata; ++sl) {
        const int slot = sl * p.pixels + pix;
        f3 c = mk3(p.b.radx[slot], p.b.rady[slot], p.b.radz[slot]);
        const int sf = p.b.sflags[slot];
        if (sf & WF_SH_UNOCCLUDED)                                         // last vertex' light sample of an ended path
            c = c + mk3(p.b.pendx[slot], p.b.pendy[slot], p.b.pendz[slot]);
        else if (sf & WF_SH_NF_MASK) c = poisonNonFinite(c, (sf >> WF_SH_NF_SHIFT) & 7);
        if (c.x > 1.0f) c.x = 1.0f;                                        // camera.cpp:110-112
        if (c.y > 1.0f) c.y = 1.0f;
        if (c.z > 1.0f) c.z = 1.0f;
        acc = acc + c;                                                     // image.hpp:82-86
    }
    p.acc[3 * px] = acc.x; p.acc[3 * px + 1] = acc.y; p.acc[3 * px + 2] = acc.z;
    if (write_img && p.img) {
        const float inv = (float) (s0 + nstrata);
        p.img[3 * px] = wfToByte(acc.x / inv); p.img[3 * px + 1] = wfToByte(acc.y / inv); p.img[3 * px + 2] = wfToByte(acc.z / inv);
    }
}

} // namespace jtx

using namespace jtx;

static size_t wfTraceLds(const DevScene &sc, bool lds, bool wide) {
    size_t b = WBLOCK * sizeof(int);
    if (lds) b += ((size_t) 2 * 8 * sc.num_nodes + (size_t) 3 * sc.num_prims) * sizeof(float4);
    if (wide) b += (size_t) sc.wide_depth * WBLOCK * sizeof(uint2);
    return b;
}

hipError_t jtx_wf_generate(const WfParams &p, int s0, int nstrata, hipStream_t st) {
    hipLaunchKernelGGL(k_wf_generate, dim3((p.num_slots + WBLOCK - 1) / WBLOCK), dim3(WBLOCK), 0, st, p, s0, nstrata);
    return hipGetLastError();
}
hipError_t jtx_wf_trace(const WfParams &p, int any, int grid, bool count, hipStream_t st) {
    const bool lds = p.scene.lds_threaded != 0;
    const bool wide = !lds && !count && p.scene.wide != nullptr;
    const size_t sh = wfTraceLds(p.scene, lds, wide);
    const dim3 g(grid), b(WBLOCK);
    if (wide) {
        if (any) hipLaunchKernelGGL((k_wf_trace<1, false, false, true>), g, b, sh, st, p);
        else hipLaunchKernelGGL((k_wf_trace<0, false, false, true>), g, b, sh, st, p);
        return hipGetLastError();
    }
#define LT(A, C, L) hipLaunchKernelGGL((k_wf_trace<A, C, L, false>), g, b, sh, st, p)
    if (any) { if (count) { if (lds) LT(1, true, true); else LT(1, true, false); } else { if (lds) LT(1, false, true); else LT(1, false, false); } }
    else     { if (count) { if (lds) LT(0, true, true); else LT(0, true, false); } else { if (lds) LT(0, false, true); else LT(0, false, false); } }
#undef LT
    return hipGetLastError();
}
hipError_t jtx_wf_shade(const WfParams &p, int grid, bool count, int typeCode, int matMask, hipStream_t st) {
#define LS(C, M) hipLaunchKernelGGL((k_wf_shade<C, M>), dim3(grid), dim3(WBLOCK), 0, st, p, typeCode)
#define LSM(M) do { if (count) LS(true, M); else LS(false, M); } while (0)
    switch (matMask) {
        case 1: LSM(1); break;
        case 2: LSM(2); break;
        case 4: LSM(4); break;
        case 8: LSM(8); break;
        default: LSM(15); break;
    }
#undef LSM
#undef LS
    return hipGetLastError();
}
hipError_t jtx_wf_resolve(const WfParams &p, int s0, int nstrata, int write_img, hipStream_t st) {
    hipLaunchKernelGGL(k_wf_resolve, dim3((p.pixels + WBLOCK - 1) / WBLOCK), dim3(WBLOCK), 0, st, p, s0, nstrata, write_img);
    return hipGetLastError();
}
