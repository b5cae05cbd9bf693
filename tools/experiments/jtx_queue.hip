// jtx_queue.hip -- k_render_queue: the timed kernel of HBM-resident scenes (8-ary BVH), RAY QUEUES PER LANE.
//
// k_render_paths gives a lane one path and runs closestHit, shading and anyHit in lock step: per traversal call a wave
// sits through ~27 node iterations for 12.35 node steps per ray (atrium) -- it waits for the longest of 64 rays, twice
// per bounce -- and the path state (beta, radiance, RNG ...) stays live across the traversal loops (91-161 spilled VGPRs).
// Here a lane owns QK paths ("slots") whose state lives in a wave-private, lane-coalesced record in HBM:
//   shade phase : slot by slot, for the lanes whose slot has its rays back -- add the shadow-ray result of the previous
//                 vertex, shade the extension-ray hit (integrateMIS, integrator.cpp:171-216, all of it: light sample, MIS
//                 weight, BSDF sample) and leave behind a shadow ray + what it adds when it is unoccluded / occluded, and
//                 the next extension ray; a slot whose path ended writes rad[stratum][pixel] and takes the next path of
//                 the wave's chunk;
//   trace phase : ONE loop in which every lane walks its own queue of up to 2 QK rays (shadow rays: any hit, slot order;
//                 extension rays: closest hit, octant order) back to back.  The loop is left for the next shade phase as
//                 soon as JTX_Q_SHADE_VOTE lanes have run out of rays -- lanes in the middle of a ray park their
//                 traversal state (stack in LDS, 2 x 16 B in the record) and resume afterwards.
// So no lane waits for another lane's ray (only for votes), nothing moves between lanes (what sank the re-entrant stream
// kernel and the wave pool), the shade phase still finds several finished slots per lane, and the
// traversal loop keeps only {origin, 1/d, t.max, group, stack pointer, queue bits} in registers.
// Every path performs the operations of k_render_paths' path in the same order (the shadow ray's contribution is
// computed before the BSDF sample but ADDED before anything later: same float sums), so the film is bit-identical.
#include "jtx_scene_dev.hpp"
#include "jtx_launch.hpp"
#include <cstdlib>

namespace jtx {

#ifndef JTX_Q_SLOTS
#define JTX_Q_SLOTS 4            // paths per lane
#endif
#ifndef JTX_Q_OCC
#define JTX_Q_OCC 8              // waves per SIMD
#endif
#ifndef JTX_Q_REFILL_VOTE
#define JTX_Q_REFILL_VOTE 8      // lanes between two rays that interrupt the walk to take their next ray
#endif
#ifndef JTX_Q_SHADE_VOTE
#define JTX_Q_SHADE_VOTE 16      // lanes out of rays that end the trace phase
#endif
#ifdef JTX_PROFILE_QUEUE
#define QSTAT(x) x
struct QStats { unsigned long long v[16]; };
#else
#define QSTAT(x)
struct QStats {};
#endif
constexpr int QK = JTX_Q_SLOTS;
constexpr int QREC = 8;                                       // float4 per slot
constexpr int QRES = QK * QREC;                               // first result record
constexpr int QRESUME = QRES + 2 * QK;                        // the two records of a parked traversal
constexpr int QWAVE_F4 = (QRESUME + 2) * 64;                  // per wave: state[QK][QREC][64] + results[2 QK][64] + resume[2][64]
static_assert(QK >= 1 && QK <= 8, "2 QK queue bits (+ 2 QK irregular-ray bits) per lane");

// slot record (float4 index f, each [64 lanes]):
//   0: o.xyz d.x | 1: d.yz beta.xy | 2: beta.z radiance.xyz | 3: rng depth stratum pixel-slot
//   4: shadow origin.xyz wi.x | 5: wi.yz tmax - | 6: add(unoccluded).xyz add(occluded).x | 7: add(occluded).yz - -
// queue entry e = 2 k: the shadow ray of slot k (records 4, 5), e = 2 k + 1: its extension ray (records 0, 1): both
// read "origin.xyz dir.x | dir.yz tmax" the same way.  results[e]: shadow {occluded ? 1 : 0}, extension {t, prim or -1, b1, b2}.
// All addressing is (wave-uniform base) + 32-bit byte offset, so that the base stays in SGPRs.
#ifndef JTX_Q_NONTEMPORAL
#define JTX_Q_NONTEMPORAL 0
#endif
typedef float qf4 __attribute__((ext_vector_type(4)));
JD float4 qLoad(const float4 *wb, unsigned off) {
#if JTX_Q_NONTEMPORAL
    const qf4 v = __builtin_nontemporal_load((const qf4 *) ((const char *) wb + off));   // streamed: the records must not evict BVH nodes
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *(const float4 *) ((const char *) wb + off);
#endif
}
JD void qStore(float4 *wb, unsigned off, float4 v) {
#if JTX_Q_NONTEMPORAL
    qf4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, (qf4 *) ((char *) wb + off));
#else
    *(float4 *) ((char *) wb + off) = v;
#endif
}
JD unsigned qRec(int rec, unsigned laneOff) { return (unsigned) rec * 1024u + laneOff; }
JD int queueRecord(int e) { return (e >> 1) * QREC + ((e & 1) ? 0 : 4); }

struct QPath { f3 o, d, beta, radiance; Rng rng; int depth; };

// integrateMIS's loop body for one extension-ray result.  true = the path has no next ray; newShadow = a shadow ray was
// left in records 4-7 (its contribution is still outstanding).
template <int MASK>
JD bool queueBounce(const DevScene &sc, int maxDepth, QPath &ps, const HitRec &h, float4 *wb, unsigned sk, bool &newShadow) {
    if (h.prim < 0) {                                                   // integrator.cpp:183-187
        ps.radiance = ps.radiance + ps.beta * a3(sc.sky);
        return true;
    }
    if (ps.depth++ == maxDepth) return true;                            // integrator.cpp:191
    const Surface sf = makeSurface(sc.shade, h, ps.o, ps.d);
    const DMaterial &mat = sc.materials[sf.material];
    ShadeCtx ctx; ctx.materials = sc.materials; ctx.textures = sc.textures; ctx.texels = sc.texels;
    const f3 wo = -ps.d;
    if (sc.num_lights > 0) {                                            // sampleLights integrator.cpp:134-169
        const uint32_t idx = ps.rng.sampleRange(sc.num_lights - 1);
        const DLight &light = sc.lights[idx];
        (void) ps.rng.f(); (void) ps.rng.f();
        LightSample ls;
        if (lightSample(light, sf.point, ls)) {
            const f3 sOrigin = sf.point + sf.normal * RAY_EPSILON;
            const float lDist = len(sf.point - ls.p);
            f3 f; float pb;
            evalPdfBxdf<MASK>(ctx, mat, sf.normal, sf.uv, wo, ls.wi, f, pb);
            f = f * absdot(ls.wi, sf.normal);
            const float pl = 1.0f / (float) sc.num_lights * ls.pdf;
            const float misWeight = powerHeuristic(1.0f, pl, 1.0f, pb);  // applied to delta lights too (Q10)
            const f3 addV = ps.beta * (misWeight * f * ls.radiance / pl);
            const f3 addO = ps.beta * mk3(0.0f);                         // integrator.cpp:168,195: beta * {} (NaN where beta overflowed)
            qStore(wb, sk + 4 * 1024, make_float4(sOrigin.x, sOrigin.y, sOrigin.z, ls.wi.x));
            qStore(wb, sk + 5 * 1024, make_float4(ls.wi.y, ls.wi.z, lDist - RAY_EPSILON, 0.0f));
            qStore(wb, sk + 6 * 1024, make_float4(addV.x, addV.y, addV.z, addO.x));
            qStore(wb, sk + 7 * 1024, make_float4(addO.y, addO.z, 0.0f, 0.0f));
            newShadow = true;
        }
    }
    const float u = ps.rng.f();
    f2 u2; u2.x = ps.rng.f(); u2.y = ps.rng.f();
    BSample bs;
    if (!sampleBxdf<MASK>(ctx, mat, sf.normal, sf.uv, wo, u, u2, bs)) return true;
    if (bs.pdf > 0.0f) ps.beta = ps.beta * (bs.f * absdot(bs.wi, sf.normal) / bs.pdf);
    ps.o = sf.point + bs.wi * RAY_EPSILON;                               // integrator.cpp:212
    ps.d = bs.wi;
    return false;
}

// What a lane keeps in registers while it walks: the ray without its direction (re-read at a leaf), the current group
// of children, the stack pointer.  cur = queue entry under way (-1: between rays); todo = bits 0-15 entries waiting,
// bits 16-31 irregular entries (walked on the binary records after the loop).
struct QWalk {
    f3 o, inv; float tmax; int negmask;
    unsigned gbase, gbits; int sp, pendLeaf;
    bool hit;
};

JD bool queueLoadRay(const float4 *wb, unsigned laneOff, int e, QWalk &w) {
    const int rc = queueRecord(e);
    const float4 a0 = qLoad(wb, qRec(rc, laneOff)), a1 = qLoad(wb, qRec(rc + 1, laneOff));
    w.o = mk3(a0.x, a0.y, a0.z);
    w.inv = mk3(1.0f / a0.w, 1.0f / a1.x, 1.0f / a1.y);
    w.negmask = (w.inv.x < 0.0f ? 1 : 0) | (w.inv.y < 0.0f ? 2 : 0) | (w.inv.z < 0.0f ? 4 : 0);
    w.tmax = (e & 1) ? __builtin_inff() : a1.z;
    return wideRayOk(w.o, w.inv, (e & 1) ? 0.001f : 0.0f, w.tmax);
}

JD void queueFinish(float4 *wb, unsigned laneOff, int &cur, bool hit) {
    if (!(cur & 1)) qStore(wb, qRec(QRES + cur, laneOff), make_float4(hit ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f));
    else if (!hit) qStore(wb, qRec(QRES + cur, laneOff), make_float4(0.0f, __int_as_float(-1), 0.0f, 0.0f));
    cur = -1;
}

// irregular rays (a zero / non-finite direction component ...): the reference's own walk on the binary records
JD void queueTraceIrregular(const DevScene &sc, const WideSrc &src, float4 *wb, unsigned laneOff, unsigned &todo) {
    while (__ballot((todo >> 16) != 0u) != 0ull) {
        if ((todo >> 16) != 0u) {
            const int e = __builtin_ctz(todo >> 16); todo &= ~(1u << (16 + e));
            const int rc = queueRecord(e);
            const float4 a0 = qLoad(wb, qRec(rc, laneOff)), a1 = qLoad(wb, qRec(rc + 1, laneOff));
            const f3 o = mk3(a0.x, a0.y, a0.z), d = mk3(a0.w, a1.x, a1.y);
            const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            const int negmask = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
            Counters9 cnt = {};
            HitRec h; h.t = 0.0f; h.prim = -1; h.b1 = 0.0f; h.b2 = 0.0f;
            if ((e & 1) == 0) {
                const bool occ = traverseThreaded<true, false, false>(src, sc.num_nodes, o, d, inv, negmask, 0.0f, a1.z, h, cnt);
                qStore(wb, qRec(QRES + e, laneOff), make_float4(occ ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f));
            } else {
                const bool hit = traverseThreaded<false, false, false>(src, sc.num_nodes, o, d, inv, negmask, 0.001f, __builtin_inff(), h, cnt);
                qStore(wb, qRec(QRES + e, laneOff), make_float4(h.t, __int_as_float(hit ? h.prim : -1), h.b1, h.b2));
            }
        }
    }
}

// Every lane walks the entries of `todo` one after the other; returns when JTX_Q_SHADE_VOTE lanes have nothing left to
// walk (canShade: those lanes have something for the shade phase) or nobody walks any more.
// OUT OF LINE on purpose: inlined into the kernel, the register allocator spilled the loop's own control variables
// (cur, todo: reloaded at every iteration) to keep shade-phase values in registers -- 2 x slower.  As a function the loop
// has its own allocation; the uniform pointers are made scalar again with readfirstlane.
typedef __attribute__((address_space(3))) unsigned long long QLdsEntry;   // a stack entry {gbase, gbits} in LDS
struct QTraceOut { unsigned todo; int cur; };
template <class T> JD T *qUniform(T *p) {
    const unsigned long long v = (unsigned long long) p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) v), hi = __builtin_amdgcn_readfirstlane((unsigned) (v >> 32));
    return (T *) (((unsigned long long) hi << 32) | lo);
}
#ifdef JTX_PROFILE_QUEUE
#define QS_PARAM , QStats &qs
#define QS_ARG , qs
#else
#define QS_PARAM
#define QS_ARG
#endif
__device__ __attribute__((noinline)) QTraceOut queueTrace(const uint4 *wide_, const float4 *tris_, float4 *wb_, unsigned laneOff, QLdsEntry *stk,
                                                          unsigned todo, int cur, bool canShade QS_PARAM) {
    const char *__restrict__ wide = (const char *) qUniform(wide_);
    float4 *wb = qUniform(wb_);
    WideSrc src; src.wide = nullptr; src.tnodes = nullptr; src.tris = qUniform(tris_); src.stk = nullptr; src.stride = 64;
    const unsigned long long shadeable = __ballot(canShade);             // wave-uniform: no register per lane
    QWalk w; w.o = mk3(0.0f); w.inv = mk3(1.0f); w.tmax = 0.0f; w.negmask = 0; w.gbase = 0u; w.gbits = 0u; w.sp = 0; w.pendLeaf = -1; w.hit = false;
    if (cur >= 0) {                                                       // resume the ray parked before the shade phase
        (void) queueLoadRay(wb, laneOff, cur, w);
        const float4 r0 = qLoad(wb, qRec(QRESUME, laneOff)), r1 = qLoad(wb, qRec(QRESUME + 1, laneOff));
        w.gbase = __float_as_uint(r0.x); w.gbits = __float_as_uint(r0.y); w.sp = __float_as_int(r0.z); w.pendLeaf = __float_as_int(r0.w);
        w.tmax = r1.x; w.hit = __float_as_int(r1.y) != 0;
    }
    while (true) {
        const bool mine = cur >= 0;
        const unsigned long long walking = __ballot(mine && w.pendLeaf < 0);
        const unsigned long long parked = __ballot(mine && w.pendLeaf >= 0);
        const unsigned long long takers = __ballot(!mine && (todo & 0xffffu) != 0u);
        const unsigned long long busy = walking | parked;
        if (takers != 0ull && (busy == 0ull || __popcll(takers) >= JTX_Q_REFILL_VOTE)) {
            QSTAT(qs.v[7]++;)
            if (!mine && (todo & 0xffffu) != 0u) {
                const int e = __builtin_ctz(todo & 0xffffu); todo &= ~(1u << e);
                QSTAT(qs.v[8]++;)
                if (queueLoadRay(wb, laneOff, e, w)) { cur = e; w.gbase = 1u << 28; w.gbits = 1u; w.sp = 0; w.pendLeaf = -1; w.hit = false; }
                else todo |= 1u << (16 + e);                              // irregular ray: after the loop
            }
            continue;
        }
        if (busy == 0ull) break;
        if (__popcll(__ballot(!mine && (todo & 0xffffu) == 0u) & shadeable) >= JTX_Q_SHADE_VOTE) break;   // lanes out of rays (takers wait for their vote)
        const int nw = __popcll(walking), np = __popcll(parked);
        if (np != 0 && (nw == 0 || np >= JTX_WIDE_LEAF_VOTE || nw <= JTX_WIDE_FEW_WALKERS)) {
            // ---- leaf step: AABB::hit on the exact box, then the leaf's triangles (mesh.hpp:106-192) ----
            QSTAT(qs.v[5]++; if (mine && w.pendLeaf >= 0) qs.v[6]++;)
            if (mine && w.pendLeaf >= 0) {
                const bool any = !(cur & 1);
                const float tmin = any ? 0.0f : 0.001f;
                const uint4 ua = *(const uint4 *) (wide + ((unsigned) w.pendLeaf << 4)), ub = *(const uint4 *) (wide + ((unsigned) w.pendLeaf << 4) + 16);
                const int rc = queueRecord(cur);
                const float4 a0 = qLoad(wb, qRec(rc, laneOff)), a1 = qLoad(wb, qRec(rc + 1, laneOff));
                const f3 d = mk3(a0.w, a1.x, a1.y);
                const float4 la = make_float4(__uint_as_float(ua.x), __uint_as_float(ua.y), __uint_as_float(ua.z), __uint_as_float(ua.w));
                const float4 lb = make_float4(__uint_as_float(ub.x), __uint_as_float(ub.y), 0.0f, 0.0f);
                if (slabRegular(la, lb, w.o, w.inv, tmin, w.tmax)) {
                    const int n = (int) ub.w, off = (int) ub.z;
                    for (int i = 0; i < n; ++i) {
                        const int prim = off + i;
                        float b1, b2, root;
                        if (!triTest(src, prim, w.o, d, tmin, w.tmax, b1, b2, root)) continue;
                        w.hit = true;
                        if (any) break;
                        w.tmax = root;
                        qStore(wb, qRec(QRES + cur, laneOff), make_float4(root, __int_as_float(prim), b1, b2));
                    }
                }
                w.pendLeaf = -1;
                if (any && w.hit) queueFinish(wb, laneOff, cur, true);
            }
            continue;
        }
        // ---- node step of the walking lanes (wideNodeStep, with the ray's kind read off the entry number) ----
        QSTAT(qs.v[3]++; if (mine && w.pendLeaf < 0) qs.v[4]++; else if (mine) qs.v[11]++; else if ((todo & 0xffffu) != 0u) qs.v[10]++; else qs.v[9]++;)
        if (mine && w.pendLeaf < 0) {
            bool go = true;
            if ((w.gbits & 0xffu) == 0u) {                               // group exhausted: pop
                if (w.sp == 0) { queueFinish(wb, laneOff, cur, w.hit); go = false; }
                else { --w.sp; const unsigned long long e = stk[w.sp * 64]; w.gbase = (unsigned) e; w.gbits = (unsigned) (e >> 32); }
            }
            if (go) {
                const bool ordered = (cur & 1) != 0;
                const int k = ordered ? __builtin_ctz(w.gbits & 0xffu) : 31 - __builtin_clz(w.gbits & 0xffu);
                w.gbits &= ~(1u << k);
                const unsigned slot = (w.gbits >> (8 + 3 * k)) & 7u;
                const unsigned ni = w.gbase >> 28, base = w.gbase & 0x0fffffffu;
                if (slot >= ni) w.pendLeaf = (int) (base + WIDE_NODE_G * ni + 2u * (slot - ni));
                else {
                    const unsigned a = (base + WIDE_NODE_G * slot) << 4;
                    if (w.gbits & 0xffu) { stk[w.sp * 64] = (unsigned long long) w.gbase | ((unsigned long long) w.gbits << 32); ++w.sp; }
                    const uint4 n0 = *(const uint4 *) (wide + a), n2 = *(const uint4 *) (wide + a + 16), n3 = *(const uint4 *) (wide + a + 32),
                                n4 = *(const uint4 *) (wide + a + 48);
                    const uint4 tl = *(const uint4 *) (wide + a + 64 + ((unsigned) (w.negmask >> 2) << 4));
                    const unsigned cbase = tl.x;
                    const unsigned hits = wideNodeHits(n0, n2, n3, n4, w.o, w.inv, ordered ? 0.001f : 0.0f, w.tmax);
                    const unsigned perm = ordered ? wideOrderOf(tl, w.negmask) : 0x00fac688u;
                    unsigned pend = 0u;
#pragma unroll
                    for (int k2 = 0; k2 < 8; ++k2) pend |= wideBit(hits, wideField3(perm, 3 * k2)) << k2;
                    pend &= (1u << (n0.w >> 28)) - 1u;
                    w.gbase = cbase | (((n0.w >> 24) & 0xfu) << 28);
                    w.gbits = pend | (perm << 8);
                }
            }
        }
    }
    if (cur >= 0) {                                                       // park the ray under way
        qStore(wb, qRec(QRESUME, laneOff), make_float4(__uint_as_float(w.gbase), __uint_as_float(w.gbits), __int_as_float(w.sp), __int_as_float(w.pendLeaf)));
        qStore(wb, qRec(QRESUME + 1, laneOff), make_float4(w.tmax, __int_as_float(w.hit ? 1 : 0), 0.0f, 0.0f));
    }
    QTraceOut out; out.todo = todo; out.cur = cur;
    return out;
}

template <int MASK>
__global__ void __launch_bounds__(64, JTX_Q_OCC) k_render_queue(RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const DevScene &sc = p.scene;
    const int lane = threadIdx.x;
    const unsigned laneOff = (unsigned) lane * 16u;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int nchunks = p.num_subblocks * p.num_groups;
    float4 *wb = p.qstate + (size_t) blockIdx.x * QWAVE_F4;             // wave-uniform
    uint2 *stk = (uint2 *) smem + lane;
    WideSrc src; src.wide = sc.wide; src.tnodes = sc.tnodes; src.tris = sc.tris; src.stk = stk; src.stride = 64;

    // the wave's current chunk (wave-uniform), as in k_render_paths
    int next = 0, nunits = 0;
    int row0 = 0, col0 = 0, slot0 = 0, sBegin = 0;
    bool exhausted = false;
    // this lane's rays: fl = in flight (bit 2k shadow ray, bit 2k+1 extension ray of slot k), todo = not yet walked, cur = under way
    unsigned fl = 0u, todo = 0u;
    int cur = -1;
    QStats qs = {}; (void) qs;
    QSTAT(long long qt = clock64();)
    while (true) {
        // ---- shade phase: the slots whose rays are all back ----
#pragma unroll 1
        for (int k = 0; k < QK; ++k) {
            const unsigned sk = qRec(k * QREC, laneOff);
            const unsigned kb = 3u << (2 * k);
            const bool hadS = (fl >> (2 * k)) & 1u, hadE = (fl >> (2 * k + 1)) & 1u;
            const bool waiting = (((todo | (todo >> 16)) & kb) != 0u) || (cur >> 1) == k;
            const bool ready = (hadS || hadE) && !waiting;
            bool need = !(hadS || hadE);
            if (__ballot(ready || (need && !exhausted)) == 0ull) continue;
            QSTAT(qs.v[12]++; if (ready) qs.v[13]++;)
            if (ready) {
                QPath ps;
                const float4 a2 = qLoad(wb, sk + 2 * 1024), a3 = qLoad(wb, sk + 3 * 1024);
                ps.beta.z = a2.x; ps.radiance = mk3(a2.y, a2.z, a2.w);
                ps.rng.state = __float_as_uint(a3.x); ps.depth = __float_as_int(a3.y);
                const int s = __float_as_int(a3.z), slot = __float_as_int(a3.w);
                if (hadS) {                                              // the light sample of the previous vertex (integrator.cpp:195)
                    const float occ = qLoad(wb, qRec(QRES + 2 * k, laneOff)).x;
                    const float4 a6 = qLoad(wb, sk + 6 * 1024), a7 = qLoad(wb, sk + 7 * 1024);
                    ps.radiance = ps.radiance + (occ != 0.0f ? mk3(a6.w, a7.x, a7.y) : mk3(a6.x, a6.y, a6.z));
                }
                bool done = true, newS = false;
                if (hadE) {
                    const float4 a0 = qLoad(wb, sk), a1 = qLoad(wb, sk + 1024);
                    ps.o = mk3(a0.x, a0.y, a0.z); ps.d = mk3(a0.w, a1.x, a1.y); ps.beta.x = a1.z; ps.beta.y = a1.w;
                    const float4 rr = qLoad(wb, qRec(QRES + 2 * k + 1, laneOff));
                    HitRec h; h.t = rr.x; h.prim = __float_as_int(rr.y); h.b1 = rr.z; h.b2 = rr.w;
                    done = queueBounce<MASK>(sc, p.max_depth, ps, h, wb, sk, newS);
                }
                fl &= ~kb;
                if (newS) { fl |= 1u << (2 * k); todo |= 1u << (2 * k); }
                if (!done) {
                    fl |= 2u << (2 * k); todo |= 2u << (2 * k);
                    qStore(wb, sk, make_float4(ps.o.x, ps.o.y, ps.o.z, ps.d.x));
                    qStore(wb, sk + 1024, make_float4(ps.d.y, ps.d.z, ps.beta.x, ps.beta.y));
                }
                if (!done || newS) {
                    qStore(wb, sk + 2 * 1024, make_float4(ps.beta.z, ps.radiance.x, ps.radiance.y, ps.radiance.z));
                    qStore(wb, sk + 3 * 1024, make_float4(__uint_as_float(ps.rng.state), __int_as_float(ps.depth), a3.z, a3.w));
                } else {
                    f3 c = ps.radiance;                                  // camera.cpp:110-112
                    if (c.x > 1.0f) c.x = 1.0f;
                    if (c.y > 1.0f) c.y = 1.0f;
                    if (c.z > 1.0f) c.z = 1.0f;
                    p.rad[(size_t) (s - p.sample_begin) * p.rad_stride + slot] = make_float4(c.x, c.y, c.z, 0.0f);
                    need = true;
                }
            }
            // ---- hand out paths of the wave's chunk to the lanes whose slot k is free ----
            while (true) {
                const unsigned long long mask = __ballot(need);
                if (mask == 0ull) break;
                if (next >= nunits) {
                    if (exhausted) { need = false; break; }
                    int c = 0;
                    if (lane == 0) {
                        c = (int) atomicAdd(p.work, 1u);
                        // cancellation poll as in k_render_paths: every 64th fetch reads the host's flag
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
                        const int s = sBegin + (u >> 6);
                        Rng rng; f3 o, d;
                        rng.seed(row, col, (uint32_t) s + 1u);             // camera.cpp:101
                        cameraRay(p.cam, col, row, s, rng, o, d);
                        qStore(wb, sk, make_float4(o.x, o.y, o.z, d.x));
                        qStore(wb, sk + 1024, make_float4(d.y, d.z, 1.0f, 1.0f));
                        qStore(wb, sk + 2 * 1024, make_float4(1.0f, 0.0f, 0.0f, 0.0f));
                        qStore(wb, sk + 3 * 1024, make_float4(__uint_as_float(rng.state), __int_as_float(0), __int_as_float(s), __int_as_float(slot0 + pl)));
                        fl |= 2u << (2 * k); todo |= 2u << (2 * k);
                        need = false;
                    }
                }
            }
        }
        QSTAT({ const long long n_ = clock64(); qs.v[1] += n_ - qt; qt = n_; qs.v[0]++; })
        if (__ballot(fl != 0u) == 0ull) break;
        // ---- trace phase ----
        { const QTraceOut t = queueTrace(sc.wide, sc.tris, wb, laneOff, (QLdsEntry *) smem + lane, todo, cur, fl != 0u || !exhausted QS_ARG);
          todo = t.todo; cur = t.cur; }
        if (__ballot((todo >> 16) != 0u) != 0ull) queueTraceIrregular(sc, src, wb, laneOff, todo);
        QSTAT({ const long long n_ = clock64(); qs.v[2] += n_ - qt; qt = n_; })
    }
#ifdef JTX_PROFILE_QUEUE
    // diagnostic build (tools/queue_stats.py): wave view [0] rounds [1] shade clocks [2] trace clocks [3] node iterations
    // [5] leaf iterations [7] refill blocks [12] slot steps of the shade phase; lane sums [4] node steps [6] leaf steps [8] rays
    // [9] node iterations out of rays [10] ... waiting for a refill [11] ... parked on a leaf [13] slots shaded
    if (p.counters) for (int i = 0; i < 14; ++i) {
        const bool perWave = i <= 3 || i == 5 || i == 7 || i == 12;
        unsigned long long v = qs.v[i];
        if (!perWave) for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) atomicAdd(&p.counters[24 + i], v);
    }
#endif
}

} // namespace jtx

using namespace jtx;

size_t jtx_queue_state_float4(int num_cus) { return (size_t) num_cus * 4 * JTX_Q_OCC * QWAVE_F4; }

hipError_t jtx_launch_render_queue(const RenderParams &p, int num_owned_tiles, int num_cus, hipStream_t stream) {
    if (num_owned_tiles <= 0) return hipSuccess;
    const size_t shmem = (size_t) p.scene.wide_depth * 64 * sizeof(uint2);
    long waves = (long) num_cus * 4 * JTX_Q_OCC;
    const long chunks = (long) p.num_subblocks * p.num_groups;
    if (waves > chunks) waves = chunks;
    const dim3 grid((unsigned) waves), block(64);
    if (p.scene.material_mask == MAT_DIFFUSE_ONLY) hipLaunchKernelGGL((k_render_queue<MAT_DIFFUSE_ONLY>), grid, block, shmem, stream, p);
    else hipLaunchKernelGGL((k_render_queue<MAT_ALL>), grid, block, shmem, stream, p);
    return hipGetLastError();
}
