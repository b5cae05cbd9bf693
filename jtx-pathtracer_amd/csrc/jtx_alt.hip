// jtx_alt.hip -- k_render_alt: the reference's OTHER integrators and the features integrateMIS leaves out.
//
//   LI = 1  integrate        (integrator.cpp:54-132): next-event estimation without MIS, emission and sky only after a
//                            specular bounce (BSDFSample::isSpecular, set by DielectricBxDF alone), `while (beta)`
//   LI = 2  integrateBasic   (integrator.cpp:12-52): no light sampling, emission at every hit, `while (beta)`
//   LI = 0  integrateMIS     (integrator.cpp:171-216) with every BxDF incl. THIN_DIELECTRIC (Material type 4,
//                            ThinDielectricBxDF dielectric.hpp:163-207, which bxdf.cpp never dispatches): scenes that
//                            hold such a material render here instead of in the timed kernels
// camera.cpp:104-106 picks the integrator by (un)commenting a line; here it is render_opts.path_integrator.
// One lane owns one pixel and runs its strata in sample order (k_render_pixels' scheme), walking the reference's binary
// node records, so film AND ray counters equal the oracle's bit for bit.  These rows are about coverage, not speed.
#include "jtx_scene_dev.hpp"
#include "jtx_launch.hpp"
#include "jtx_tiles.hpp"

namespace jtx {

struct AltPath { f3 o, d, beta, radiance; Rng rng; int depth; bool specularBounce; };

// one trip of the integrator's loop; true = the path is finished
template <bool COUNT, int LI>
JD bool altBounce(const DevScene &sc, const GlobalSrc &src, int maxDepth, AltPath &ps, Counters9 &cnt) {
    if (LI != 0 && !nonzero(ps.beta)) return true;                        // while (beta), integrator.cpp:18,61
    HitRec h;
    const bool hit = traverseNoStack<false, COUNT>(src, sc.num_nodes, ps.o, ps.d, 0.001f, __builtin_inff(), h, cnt);
    if (!hit) {
        if (LI != 1 || ps.specularBounce) ps.radiance = ps.radiance + ps.beta * a3(sc.sky);      // integrator.cpp:21-24, 63-68, 183-187
        return true;
    }
    const Surface sf = makeSurface(sc.shade, h, ps.o, ps.d);
    const DMaterial &mat = sc.materials[sf.material];
    if (LI == 2 || (LI == 1 && ps.specularBounce)) ps.radiance = ps.radiance + ps.beta * a3(mat.emission);   // integrator.cpp:27, 74-76
    if (ps.depth++ == maxDepth) return true;
    ShadeCtx ctx; ctx.materials = sc.materials; ctx.textures = sc.textures; ctx.texels = sc.texels;
    const f3 wo = -ps.d;
    if (LI == 0 && sc.num_lights > 0) {                                   // sampleLights, integrator.cpp:134-169 (as pathBounce)
        const uint32_t idx = ps.rng.sampleRange(sc.num_lights - 1);
        const DLight &light = sc.lights[idx];
        (void) ps.rng.f(); (void) ps.rng.f();
        LightSample ls;
        if (lightSample(light, sf.point, ls)) {
            const f3 sOrigin = sf.point + sf.normal * RAY_EPSILON;
            const float lDist = len(sf.point - ls.p);
            HitRec dummy;
            const bool occluded = traverseNoStack<true, COUNT>(src, sc.num_nodes, sOrigin, ls.wi, 0.0f, lDist - RAY_EPSILON, dummy, cnt);
            if (!occluded) {
                f3 f; float pb;
                if (COUNT) countClass(cnt.n_eval_t, bxdfClass(mat));
                evalPdfBxdf<MAT_EVERY>(ctx, mat, sf.normal, sf.uv, wo, ls.wi, f, pb);
                f = f * absdot(ls.wi, sf.normal);
                const float pl = 1.0f / (float) sc.num_lights * ls.pdf;
                const float misWeight = powerHeuristic(1.0f, pl, 1.0f, pb);
                ps.radiance = ps.radiance + ps.beta * (misWeight * f * ls.radiance / pl);
            } else ps.radiance = ps.radiance + ps.beta * mk3(0.0f);
        }
    }
    if (LI == 1) {                                                        // integrator.cpp:84-112 (the host refuses a scene without lights)
        const uint32_t idx = ps.rng.sampleRange(sc.num_lights - 1);
        const DLight &light = sc.lights[idx];
        (void) ps.rng.f(); (void) ps.rng.f();                             // Vec2f u
        LightSample ls;
        if (lightSample(light, sf.point, ls) && ls.pdf > 0.0f) {
            f3 f; float pb;
            if (COUNT) countClass(cnt.n_eval_t, bxdfClass(mat));                  // (integrate evaluates BEFORE its shadow ray: integrator.cpp:94-101)
            evalPdfBxdf<MAT_EVERY>(ctx, mat, sf.normal, sf.uv, wo, ls.wi, f, pb);
            f = f * absdot(ls.wi, sf.normal);
            const f3 sOrigin = sf.point + sf.normal * RAY_EPSILON;
            const float lDist = len(sf.point - ls.p);
            if (nonzero(f)) {                                             // `f && !scene.anyHit(...)`: the shadow ray only for a non-zero f
                HitRec dummy;
                if (!traverseNoStack<true, COUNT>(src, sc.num_nodes, sOrigin, ls.wi, 0.0f, lDist - RAY_EPSILON, dummy, cnt))
                    ps.radiance = ps.radiance + ps.beta * f * ls.radiance / (ls.pdf * (1.0f / (float) sc.num_lights));
            }
        }
    }
    const float u = ps.rng.f();
    f2 u2; u2.x = ps.rng.f(); u2.y = ps.rng.f();
    BSample bs;
    if (COUNT) { cnt.n_shade++; countClass(cnt.n_shade_t, bxdfClass(mat)); }
    if (!sampleBxdf<MAT_EVERY>(ctx, mat, sf.normal, sf.uv, wo, u, u2, bs)) return true;
    if (LI != 0 || bs.pdf > 0.0f) ps.beta = ps.beta * (bs.f * absdot(bs.wi, sf.normal) / bs.pdf);   // unguarded in integrate / integrateBasic
    ps.specularBounce = bs.specular;                                      // integrator.cpp:126
    ps.o = sf.point + bs.wi * RAY_EPSILON;
    ps.d = bs.wi;
    return false;
}

JD unsigned char altToByte(float v) {                                     // image.hpp:9-16,47-52
    const float g = v > 0.0f ? sqrtf(v) : 0.0f;
    const float c = clampf(g, 0.0f, 0.999f);
    return (unsigned char) (int) (255.999f * c);
}

template <bool COUNT, int LI>
__global__ void __launch_bounds__(256) k_render_alt(RenderParams p) {
    const DevScene &sc = p.scene;
    const int slot = blockIdx.x * 256 + threadIdx.x;
    int row, col;
    Counters9 cnt = {};
    const bool inside = slot < p.rad_stride && slotToPixel(slot, p.tile_rank, p.tile_world, p.width, p.height, row, col);
    if (inside) {
        const size_t pix = (size_t) row * p.width + col;
        f3 acc = mk3(0.0f);
        if (p.sample_begin > 0) acc = mk3(p.acc[3 * pix], p.acc[3 * pix + 1], p.acc[3 * pix + 2]);
        GlobalSrc src; src.tnodes = sc.tnodes; src.tris = sc.tris;
        for (int s = p.sample_begin; s < p.sample_end; ++s) {
            AltPath ps;
            ps.rng.seed(row, col, (uint32_t) s + 1u);                     // camera.cpp:101
            cameraRay(p.cam, col, row, s, ps.rng, ps.o, ps.d);
            ps.beta = mk3(1.0f); ps.radiance = mk3(0.0f); ps.depth = 0; ps.specularBounce = true;
            if (COUNT) cnt.n_camera++;
            while (!altBounce<COUNT, LI>(sc, src, p.max_depth, ps, cnt)) {}
            f3 c = ps.radiance;                                           // camera.cpp:110-112
            if (c.x > 1.0f) c.x = 1.0f;
            if (c.y > 1.0f) c.y = 1.0f;
            if (c.z > 1.0f) c.z = 1.0f;
            acc = acc + c;                                                // image.hpp:82-86
        }
        p.acc[3 * pix] = acc.x; p.acc[3 * pix + 1] = acc.y; p.acc[3 * pix + 2] = acc.z;
        if (p.img) {
            const float inv = (float) p.sample_end;
            p.img[3 * pix] = altToByte(acc.x / inv); p.img[3 * pix + 1] = altToByte(acc.y / inv); p.img[3 * pix + 2] = altToByte(acc.z / inv);
        }
    }
    if (COUNT) {
        const unsigned v[9] = {cnt.n_camera, cnt.n_closest, cnt.n_any, cnt.n_nodes_closest, cnt.n_tri_closest, cnt.n_accept,
                               cnt.n_nodes_any, cnt.n_tri_any, cnt.n_shade};
        for (int i = 0; i < 9; ++i) {
            unsigned long long sv = v[i];
            for (int off = 32; off > 0; off >>= 1) sv += __shfl_down(sv, off, 64);
            if ((threadIdx.x & 63) == 0 && sv) atomicAdd(&p.counters[i], sv);
        }
        const unsigned w[14] = {cnt.n_shade_t[0], cnt.n_shade_t[1], cnt.n_shade_t[2], cnt.n_shade_t[3], cnt.n_shade_t[4], cnt.n_shade_t[5], cnt.n_shade_t[6],
                                cnt.n_eval_t[0], cnt.n_eval_t[1], cnt.n_eval_t[2], cnt.n_eval_t[3], cnt.n_eval_t[4], cnt.n_eval_t[5], cnt.n_eval_t[6]};
        for (int i = 0; i < 14; ++i) {
            unsigned long long sv = w[i];
            for (int off = 32; off > 0; off >>= 1) sv += __shfl_down(sv, off, 64);
            if ((threadIdx.x & 63) == 0 && sv) atomicAdd(&p.counters[i < 7 ? CNT_SHADE_T + i : CNT_EVAL_T + i - 7], sv);
        }
    }
}

// one (row, col, sample) per lane: the per-sample radiance of the chosen integrator (parity tests)
template <int LI>
__global__ void __launch_bounds__(256) k_radiance_samples_alt(DevScene sc, DCam cam, int maxDepth, int n, const int *row, const int *col,
                                                              const int *sample, float *rgb) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Counters9 cnt = {};
    AltPath ps;
    ps.rng.seed(row[i], col[i], (uint32_t) sample[i] + 1u);
    cameraRay(cam, col[i], row[i], sample[i], ps.rng, ps.o, ps.d);
    ps.beta = mk3(1.0f); ps.radiance = mk3(0.0f); ps.depth = 0; ps.specularBounce = true;
    GlobalSrc src; src.tnodes = sc.tnodes; src.tris = sc.tris;
    while (!altBounce<false, LI>(sc, src, maxDepth, ps, cnt)) {}
    f3 c = ps.radiance;
    if (c.x > 1.0f) c.x = 1.0f;
    if (c.y > 1.0f) c.y = 1.0f;
    if (c.z > 1.0f) c.z = 1.0f;
    rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
}

} // namespace jtx

using namespace jtx;

hipError_t jtx_launch_render_alt(const RenderParams &p, int num_owned_tiles, bool count, int li, hipStream_t stream) {
    if (num_owned_tiles <= 0) return hipSuccess;
    RenderParams q = p;
    q.rad_stride = num_owned_tiles * 1024;                               // slots of this shard
    const dim3 grid((unsigned) (q.rad_stride / 256)), block(256);
#define LAUNCH_ALT(C, L) hipLaunchKernelGGL((k_render_alt<C, L>), grid, block, 0, stream, q)
    if (li == 1)      { if (count) LAUNCH_ALT(true, 1); else LAUNCH_ALT(false, 1); }
    else if (li == 2) { if (count) LAUNCH_ALT(true, 2); else LAUNCH_ALT(false, 2); }
    else              { if (count) LAUNCH_ALT(true, 0); else LAUNCH_ALT(false, 0); }
#undef LAUNCH_ALT
    return hipGetLastError();
}

hipError_t jtx_launch_radiance_samples_alt(const DevScene &sc, const DCam &cam, int maxDepth, int li, int n, const int *row, const int *col,
                                           const int *sample, float *rgb, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const dim3 grid((unsigned) ((n + 255) / 256)), block(256);
    if (li == 1)      hipLaunchKernelGGL((k_radiance_samples_alt<1>), grid, block, 0, stream, sc, cam, maxDepth, n, row, col, sample, rgb);
    else if (li == 2) hipLaunchKernelGGL((k_radiance_samples_alt<2>), grid, block, 0, stream, sc, cam, maxDepth, n, row, col, sample, rgb);
    else              hipLaunchKernelGGL((k_radiance_samples_alt<0>), grid, block, 0, stream, sc, cam, maxDepth, n, row, col, sample, rgb);
    return hipGetLastError();
}
