// jtx_profile.hpp -- the diagnostic builds of the render kernels, in one place.  Every -DJTX_PROFILE_* switch of the library is defined
// here as a set of hooks that expand to NOTHING in the product build; the kernels (jtx_kernels.hip) and traversals (jtx_scene_dev.hpp)
// only name the hooks.  The tallies land in the scene's 64-word counter block behind the nine ray counters and are read back by the
// jtx_mi_debug_* entry points of the same builds (jtx_capi.hip; tools/tools_util.py, tools_wide_stats.py, tools_phases.py, tools_timeline.py).
//   JTX_PROFILE_UTIL      threaded (binary) traversal: loop iterations a wave sits through, by number of walking lanes; idle lane-iterations
//   JTX_PROFILE_WIDE      8-ary traversal: calls, node / leaf iterations and steps, triangle tests, the same histogram and idle shares
//   JTX_PROFILE_PHASES    wave clocks per phase of a bounce as lane 0 sees them (closest, light sample, shadow, BxDF, hand-out)
//   JTX_PROFILE_TIMELINE  wall-clock start / end of every persistent wave, loop iterations and live lanes
#pragma once

// ---- extra per-lane tallies (members of Counters9) ----
#ifdef JTX_PROFILE_UTIL
#define JTX_PROF_UTIL_FIELDS                                                                                             \
    unsigned it_interior, it_leaf, it_calls;   /* loop iterations this lane sat through (= wave iterations) */            \
    unsigned it_hist[7];   /* interior iterations by number of walking lanes: 1-2, 3-4, 5-8, 9-16, 17-32, 33-48, 49-64 */ \
    unsigned it_np, it_nd, it_lw, it_ld;   /* lane-iterations idle: interior iterations spent parked / done, leaf phases spent walking / done */
#define UTIL(x) x
#else
#define JTX_PROF_UTIL_FIELDS
#define UTIL(x)
#endif
#ifdef JTX_PROFILE_WIDE
#define JTX_PROF_WIDE_FIELDS                                                                                             \
    unsigned w_calls, w_node_iters, w_node_steps, w_leaf_iters, w_leaf_steps, w_tris, w_pops, w_fetch;                    \
    unsigned w_hist[7];    /* node iterations by number of walking lanes: 1-2, 3-4, 5-8, 9-16, 17-32, 33-48, 49-64 */     \
    unsigned w_np, w_nd, w_lw, w_ld;   /* lane-iterations idle: node iterations spent parked / done, leaf iterations spent walking / done */
#define WSTAT(x) x
#else
#define JTX_PROF_WIDE_FIELDS
#define WSTAT(x)
#endif

// ---- phase clocks (members of PathState; PH_DECL / PH(i) inside pathBounce) ----
#ifdef JTX_PROFILE_PHASES
#define JTX_PROF_PATH_FIELDS long long ph[6];
#define PH_DECL long long ph_t = clock64();
#define PH(i) { long long n_ = clock64(); ps.ph[i] += n_ - ph_t; ph_t = n_; }
#define JTX_PROF_PHASES_BEGIN(ps) for (int i_ = 0; i_ < 6; ++i_) (ps).ph[i_] = 0; const long long prof_k0 = clock64();
#define JTX_PROF_HANDOUT_BEGIN const long long prof_h0 = clock64();
#define JTX_PROF_HANDOUT_END(ps) (ps).ph[5] += clock64() - prof_h0;            /* the hand-out (chunk fetches, camera rays) */
#define JTX_PROF_PHASES_END(p, ps, lane, withHandout)                                                                    \
    if ((p).counters && (lane) == 0) {     /* per-phase wave clocks as lane 0 sees them */                                \
        for (int i_ = 0; i_ < 5; ++i_) atomicAdd(&(p).counters[16 + i_], (unsigned long long) (ps).ph[i_]);               \
        atomicAdd(&(p).counters[16 + 5], (unsigned long long) (clock64() - prof_k0));                                     \
        if (withHandout) atomicAdd(&(p).counters[16 + 6], (unsigned long long) (ps).ph[5]);                               \
    }
#else
#define JTX_PROF_PATH_FIELDS
#define PH_DECL
#define PH(i)
#define JTX_PROF_PHASES_BEGIN(ps)
#define JTX_PROF_HANDOUT_BEGIN
#define JTX_PROF_HANDOUT_END(ps)
#define JTX_PROF_PHASES_END(p, ps, lane, withHandout)
#endif

// ---- wave timeline ----
#ifdef JTX_PROFILE_TIMELINE
#define JTX_PROF_TIMELINE_BEGIN const long long prof_tl0 = wall_clock64(); unsigned prof_tl_iters = 0, prof_tl_active = 0; (void) prof_tl_iters; (void) prof_tl_active;
#define JTX_PROF_TIMELINE_ITER(alive) prof_tl_iters++; prof_tl_active += (alive) ? 1 : 0;
#define JTX_PROF_TIMELINE_WAVE(p, wid)                                                                                   \
    if ((wid) < 65536) { (p).counters[64 + 2 * (wid)] = (unsigned long long) prof_tl0; (p).counters[64 + 2 * (wid) + 1] = (unsigned long long) wall_clock64(); }
#define JTX_PROF_TIMELINE_END(p, lane, wid)                                                                              \
    if ((p).counters) {                                                                                                  \
        unsigned long long a_ = prof_tl_active;                                                                          \
        for (int off_ = 32; off_ > 0; off_ >>= 1) a_ += __shfl_down(a_, off_, 64);                                       \
        if ((lane) == 0) {                                                                                               \
            JTX_PROF_TIMELINE_WAVE(p, wid)                                                                               \
            atomicAdd(&(p).counters[40], (unsigned long long) prof_tl_iters); atomicAdd(&(p).counters[41], a_);          \
        }                                                                                                                \
    }
#else
#define JTX_PROF_TIMELINE_BEGIN
#define JTX_PROF_TIMELINE_ITER(alive)
#define JTX_PROF_TIMELINE_WAVE(p, wid)
#define JTX_PROF_TIMELINE_END(p, lane, wid)
#endif

// ---- exports at the end of a kernel (P = RenderParams, C = Counters9; all lanes of the wave take part) ----
#ifdef JTX_PROFILE_WIDE
// lane sums of the steps, wave maxima of the iterations (tools/tools_wide_stats.py)
template <class P, class C>
JD void profExportWide(const P &p, const C &cnt) {
    if (!p.counters) return;
    const unsigned v[12] = {cnt.w_calls, cnt.w_node_iters, cnt.w_node_steps, cnt.w_leaf_iters, cnt.w_leaf_steps, cnt.w_tris, cnt.w_pops, cnt.w_fetch,
                            cnt.w_np, cnt.w_nd, cnt.w_lw, cnt.w_ld};
    for (int i = 0; i < 12; ++i) {
        const bool perWave = (i == 0 || i == 1 || i == 3);
        unsigned long long sv = v[i];
        if (perWave) { for (int off = 32; off > 0; off >>= 1) { unsigned long long o2 = __shfl_down(sv, off, 64); sv = sv > o2 ? sv : o2; } }
        else for (int off = 32; off > 0; off >>= 1) sv += __shfl_down(sv, off, 64);
        if ((threadIdx.x & 63) == 0 && sv) atomicAdd(&p.counters[i < 8 ? 24 + i : 40 + i], sv);
    }
    if ((threadIdx.x & 63) == 0) for (int i = 0; i < 7; ++i) atomicAdd(&p.counters[9 + i], (unsigned long long) cnt.w_hist[i]);
}
#define JTX_PROF_WIDE_EXPORT(p, cnt) profExportWide(p, cnt);
#else
#define JTX_PROF_WIDE_EXPORT(p, cnt)
#endif
#ifdef JTX_PROFILE_UTIL
// every lane of a wave sits through the same traversal-loop iterations, but lanes that left the pixel loop early stop counting: wave maxima
template <class P, class C>
JD void profExportUtil(const P &p, const C &cnt) {
    unsigned a = cnt.it_interior, b = cnt.it_leaf, c = cnt.it_calls;
    for (int off = 32; off > 0; off >>= 1) { a = max(a, __shfl_down(a, off, 64)); b = max(b, __shfl_down(b, off, 64)); c = max(c, __shfl_down(c, off, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&p.counters[20], (unsigned long long) a); atomicAdd(&p.counters[21], (unsigned long long) b); atomicAdd(&p.counters[22], (unsigned long long) c);
                                   for (int i = 0; i < 7; ++i) atomicAdd(&p.counters[24 + i], (unsigned long long) cnt.it_hist[i]); }
    unsigned long long id[4] = {cnt.it_np, cnt.it_nd, cnt.it_lw, cnt.it_ld};
    for (int i = 0; i < 4; ++i) { for (int off = 32; off > 0; off >>= 1) id[i] += __shfl_down(id[i], off, 64);
                                  if ((threadIdx.x & 63) == 0) atomicAdd(&p.counters[48 + i], id[i]); }
}
#define JTX_PROF_UTIL_EXPORT(p, cnt) profExportUtil(p, cnt);
#else
#define JTX_PROF_UTIL_EXPORT(p, cnt)
#endif
