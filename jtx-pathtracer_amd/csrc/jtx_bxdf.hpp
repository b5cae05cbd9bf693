// jtx_bxdf.hpp -- device BxDF evaluation for the shading stage: Lambert, GGX conductor, GGX
// dielectric and glTF metallic-roughness, plus the sample/eval/pdf dispatch.
// Replaces src/bsdf/{bxdf.cpp,bxdf.hpp,diffuse.hpp,microfacet.hpp,conductor.hpp,dielectric.hpp,gltf.hpp}
// and util/complex.hpp of the reference; line cites are on each function.
#pragma once
#include "jtx_device_math.hpp"

namespace jtx {

// Device material record (80 B, one per material; read through the scalar/L1 path).
struct DMaterial {
    int   type;            // 0 DIFFUSE 1 DIELECTRIC 2 CONDUCTOR 3 METALLIC_ROUGHNESS (material.hpp:6-11), 4 THIN_DIELECTRIC (dielectric.hpp:163-207)
    float albedo[3];
    float ior[3];
    float k[3];
    float alpha_x, alpha_y;
    int   albedo_tex, mr_tex;
    float emission[3];     // Material::emission (material.hpp:19): read by integrate / integrateBasic only
    int   pad[3];
};
// Device texture descriptor.  `linear_off` points at texels already run through sRGBToLinear on the
// host at scene_create (color.hpp:13-23 is a pure per-texel function, so decoding once is identical
// to decoding at every lookup, bxdf.cpp:18,45); `raw_off` at the untouched texels.
struct DTexture { int w, h, c; int pad; long long raw_off; long long linear_off; };

struct BSample { f3 f; f3 wi; float pdf; bool specular; };   // specular = BSDFSample::isSpecular (bxdf.hpp:123): only DielectricBxDF sets it

// ---- Fresnel & helpers (bxdf.hpp:7-116) ----
JD f3 reflect(f3 wo, f3 n) { return -wo + 2.0f * dot(wo, n) * n; }

JD bool refract(f3 wi, f3 n, float eta, float &etap, f3 &wt) {                    // bxdf.hpp:20-41
    float ci = dot(wi, n);
    if (ci < 0.0f) { eta = 1.0f / eta; ci = -ci; n = -n; }
    etap = eta;
    float radicand = fmax2(0.0f, 1.0f - sqr(ci)) / (eta * eta);
    if (radicand >= 1.0f) return false;
    float ct = safeSqrt(1.0f - radicand);
    wt = -wi / eta + (ci / eta - ct) * n;
    return true;
}
JD f3 schlick(f3 wo, f3 wm, f3 R) {                                               // bxdf.hpp:43-48
    float c = absdot(wo, wm);
    float m = 1.0f - c;
    float m2 = m * m;
    return R + (1.0f - R) * m2 * m2 * m;
}
JD float fresnelDielectric(float ci, float eta) {                                 // bxdf.hpp:56-77
    ci = clampf(ci, -1.0f, 1.0f);
    if (ci < 0.0f) { eta = 1.0f / eta; ci = -ci; }
    float radicand = (1.0f - ci * ci) / (eta * eta);
    if (radicand >= 1.0f) return 1.0f;
    float ct = safeSqrt(1.0f - radicand);
    float r_par  = (eta * ci - ct) / (eta * ci + ct);
    float r_perp = (ci - eta * ct) / (ci + eta * ct);
    return (r_par * r_par + r_perp * r_perp) / 2.0f;
}
struct Cx { float r, i; };
JD Cx cx(float r, float i) { Cx c; c.r = r; c.i = i; return c; }
JD Cx cmul(Cx a, Cx b) { return cx(a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r); }   // complex.hpp:26-28
JD Cx cdiv(Cx a, Cx c) {                                                           // complex.hpp:30-33
    float scale = 1.0f / (c.r * c.r + c.i * c.i);
    return cx((a.r * c.r + a.i * c.i) * scale, (a.i * c.r - a.r * c.i) * scale);
}
JD float cnorm(Cx c) { return c.r * c.r + c.i * c.i; }
JD Cx csqrt(Cx c) {                                                                // complex.hpp:49-57
    float n = sqrtf(cnorm(c));
    float t1 = sqrtf(0.5f * (n + fabsf(c.r)));
    float t2 = 0.5f * c.i / t1;
    if (n == 0.0f) return cx(0.0f, 0.0f);
    if (c.r >= 0.0f) return cx(t1, t2);
    return cx(fabsf(t2), copysignf(t1, c.i));
}
JD float fresnelComplex(float ci, Cx eta) {                                        // bxdf.hpp:85-100
    ci = clampf(ci, 0.0f, 1.0f);
    float numerator = 1.0f - ci * ci;
    Cx radicand = cdiv(cx(numerator, 0.0f), cmul(eta, eta));
    Cx ct = csqrt(cx(1.0f - radicand.r, -radicand.i));
    Cx eci = cmul(eta, cx(ci, 0.0f));
    Cx r_par = cdiv(cx(eci.r - ct.r, eci.i - ct.i), cx(eci.r + ct.r, eci.i + ct.i));
    Cx ect = cmul(eta, ct);
    Cx r_perp = cdiv(cx(ci - ect.r, -ect.i), cx(ci + ect.r, ect.i));
    return (cnorm(r_par) + cnorm(r_perp)) / 2.0f;
}
JD f3 fresnelComplexRGB(float c, f3 eta, f3 k) {                                   // bxdf.hpp:110-116
    return mk3(fresnelComplex(c, cx(eta.x, k.x)), fresnelComplex(c, cx(eta.y, k.y)), fresnelComplex(c, cx(eta.z, k.z)));
}

// ---- Trowbridge-Reitz (microfacet.hpp:12-111) ----
struct GGX {
    float ax, ay;
    JD bool smooth() const { return fmax2(ax, ay) < 1e-3f; }
    JD float D(f3 wm) const {
        float t2 = tan2Theta(wm);
        if (isinf_(t2)) return 0.0f;
        float cos4 = sqr(cos2Theta(wm));
        if (cos4 < 1e-6f) return 0.0f;
        float e = t2 * (sqr(cosPhi(wm) / ax) + sqr(sinPhi(wm) / ay));
        return 1.0f / (PI_F * ax * ay * cos4 * sqr(1.0f + e));
    }
    JD float lambda(f3 w) const {
        float t2 = tan2Theta(w);
        if (isinf_(t2)) return 0.0f;
        float alpha2 = sqr(ax * cosPhi(w)) + sqr(ay * sinPhi(w));
        return 0.5f * (sqrtf(1.0f + alpha2 * t2) - 1.0f);
    }
    JD float G1(f3 w) const { return 1.0f / (1.0f + lambda(w)); }
    JD float G(f3 wo, f3 wi) const { return 1.0f / (1.0f + lambda(wo) + lambda(wi)); }
    JD float pdf(f3 w, f3 wm) const { return G1(w) / absCosTheta(w) * D(wm) * absdot(w, wm); }
    JD f3 sampleWm(f3 w, f2 u) const {                                             // microfacet.hpp:86-106
        f3 wh = normalize(mk3(ax * w.x, ay * w.y, w.z));
        if (wh.z < 0.0f) wh = -wh;
        f3 t1 = (wh.z < 0.99999f) ? normalize(cross(mk3(0.0f, 0.0f, 1.0f), wh)) : mk3(1.0f, 0.0f, 0.0f);
        f3 t2 = cross(wh, t1);
        f2 p = sampleUniformDiskPolar(u);
        float h = sqrtf(1.0f - sqr(p.x));
        p.y = lerpf(h, p.y, (1.0f + wh.z) / 2.0f);
        float pz = sqrtf(fmax2(0.0f, 1.0f - (p.x * p.x + p.y * p.y)));
        f3 nh = p.x * t1 + p.y * t2 + pz * wh;
        return normalize(mk3(ax * nh.x, ay * nh.y, fmax2(1e-6f, nh.z)));
    }
};

// ---- Lambert (diffuse.hpp:5-29) ----
JD f3 diffuseEval(f3 R, f3 wo, f3 wi) { return sameHemisphere(wo, wi) ? R * INV_PI : mk3(0.0f); }
JD bool diffuseSample(f3 R, f3 wo, f2 u, BSample &s) {
    f3 wi = sampleCosineHemisphere(u);
    if (wo.z < 0.0f) wi.z *= -1.0f;
    s.f = R * INV_PI; s.wi = wi; s.pdf = cosineHemispherePDF(absCosTheta(wi));
    return true;
}
JD float diffusePdf(f3 wo, f3 wi) { return sameHemisphere(wo, wi) ? cosineHemispherePDF(absCosTheta(wi)) : 0.0f; }

// ---- Conductor (conductor.hpp:6-74) ----
JD f3 conductorEval(GGX mf, f3 eta, f3 k, f3 wo, f3 wi) {
    if (mf.smooth()) return mk3(0.0f);
    float co = absCosTheta(wo), ci = absCosTheta(wi);
    if (co == 0.0f || ci == 0.0f) return mk3(0.0f);
    f3 wm = wi + wo;
    if (lenSqr(wm) == 0.0f) return mk3(0.0f);
    wm = normalize(wm);
    f3 F = fresnelComplexRGB(absdot(wo, wm), eta, k);
    return mf.D(wm) * F * mf.G(wo, wi) / (4.0f * ci * co);
}
JD bool conductorSample(GGX mf, f3 eta, f3 k, f3 wo, f2 u, BSample &s) {
    if (mf.smooth()) {
        f3 wi = mk3(-wo.x, -wo.y, wo.z);
        float ci = absCosTheta(wi);
        s.f = fresnelComplexRGB(ci, eta, k) / ci; s.wi = wi; s.pdf = 1.0f;
        return true;
    }
    if (wo.z == 0.0f) return false;
    f3 wm = mf.sampleWm(wo, u);
    f3 wi = reflect(wo, wm);
    if (!sameHemisphere(wo, wi)) return false;
    float co = absCosTheta(wo), ci = absCosTheta(wi);
    if (co == 0.0f || ci == 0.0f) return false;
    float pdf = mf.pdf(wo, wm) / (4.0f * absdot(wo, wm));
    f3 F = fresnelComplexRGB(absdot(wo, wm), eta, k);
    s.f = mf.D(wm) * F * mf.G(wo, wi) / (4.0f * ci * co); s.wi = wi; s.pdf = pdf;
    return true;
}
JD float conductorPdf(GGX mf, f3 wo, f3 wi) {
    if (mf.smooth()) return 0.0f;
    if (!sameHemisphere(wo, wi)) return 0.0f;
    f3 wm = wo + wi;
    if (lenSqr(wm) == 0.0f) return 0.0f;
    wm = faceForward(normalize(wm), mk3(0.0f, 0.0f, 1.0f));
    return mf.pdf(wo, wm) / (4.0f * absdot(wo, wm));
}

// ---- Dielectric (dielectric.hpp:5-161) ----
JD f3 dielectricEval(GGX mf, float eta, f3 wo, f3 wi) {
    if (eta == 1.0f || mf.smooth()) return mk3(0.0f);
    float co = absCosTheta(wo), ci = absCosTheta(wi);
    bool refl = co * ci > 0.0f;
    float etap = 1.0f;
    if (!refl) etap = co > 0.0f ? eta : 1.0f / eta;
    f3 wm = wi * etap + wo;
    if (ci == 0.0f || co == 0.0f || lenSqr(wm) == 0.0f) return mk3(0.0f);
    wm = faceForward(normalize(wm), mk3(0.0f, 0.0f, 1.0f));
    if (dot(wm, wi) * ci < 0.0f || dot(wm, wo) * co < 0.0f) return mk3(0.0f);
    float F = fresnelDielectric(dot(wo, wm), eta);
    if (refl) return mk3(mf.D(wm) * F * mf.G(wo, wi) / fabsf(4.0f * ci * co));
    float a = mf.D(wm) * (1.0f - F) * mf.G(wo, wi) * fabsf(dot(wi, wm) * dot(wo, wm));
    float b = sqr(dot(wi, wm) + dot(wo, wm) / etap) * fabsf(ci * co);
    return mk3(a / b);
}
JD bool dielectricSample(GGX mf, float eta, f3 wo, float uc, f2 u, BSample &s) {
    s.specular = mf.smooth();                                                      // dielectric.hpp:43, carried by every sample it returns
    if (eta == 1.0f || mf.smooth()) {
        float R = fresnelDielectric(wo.z, eta);
        float T = 1.0f - R;
        float p = R / (R + T);
        if (uc < p) {
            f3 wi = mk3(-wo.x, -wo.y, wo.z);
            s.f = mk3(R / absCosTheta(wi)); s.wi = wi; s.pdf = p;
            return true;
        }
        f3 wi; float etap;
        if (!refract(wo, mk3(0.0f, 0.0f, 1.0f), eta, etap, wi)) return false;
        s.f = mk3(T / absCosTheta(wi)); s.wi = wi; s.pdf = 1.0f - p;
        return true;
    }
    f3 wm = mf.sampleWm(wo, u);
    float R = fresnelDielectric(dot(wo, wm), eta);
    float T = 1.0f - R;
    float p = R / (R + T);
    if (uc < p) {
        f3 wi = reflect(wo, wm);
        if (!sameHemisphere(wo, wi)) return false;
        float pdf = mf.pdf(wo, wm) / (4.0f * absdot(wo, wm)) * p;
        float f = mf.D(wm) * mf.G(wo, wi) * R / (4.0f * absCosTheta(wi) * absCosTheta(wo));
        s.f = mk3(f); s.wi = wi; s.pdf = pdf;
        return true;
    }
    float etap; f3 wi = mk3(0.0f);
    bool tir = !refract(wo, wm, eta, etap, wi);
    if (sameHemisphere(wo, wi) || wi.z == 0.0f || tir) return false;
    float dn = absdot(wi, wm) / sqr(dot(wi, wm) + dot(wo, wm) / etap);
    float pdf = mf.pdf(wo, wm) * dn * (1.0f - p);
    float f = mf.D(wm) * T * mf.G(wo, wi) * fabsf(dot(wi, wm) * dot(wo, wm));
    f /= sqr(dot(wi, wm) + dot(wm, wo) / etap) * fabsf(wi.z * wo.z);
    s.f = mk3(f); s.wi = wi; s.pdf = pdf;
    return true;
}
JD float dielectricPdf(GGX mf, float eta, f3 wo, f3 wi) {
    if (eta == 1.0f || mf.smooth()) return 0.0f;
    float co = absCosTheta(wo), ci = absCosTheta(wi);
    bool refl = co * ci > 0.0f;
    float etap = 1.0f;
    if (!refl) etap = co > 0.0f ? eta : 1.0f / eta;
    f3 wm = wi * etap + wo;
    if (ci == 0.0f || co == 0.0f || lenSqr(wm) == 0.0f) return 0.0f;
    wm = faceForward(normalize(wm), mk3(0.0f, 0.0f, 1.0f));
    if (dot(wm, wi) * ci < 0.0f || dot(wm, wo) * co < 0.0f) return 0.0f;
    float R = fresnelDielectric(dot(wo, wm), eta);
    float T = 1.0f - R;
    if (refl) return mf.pdf(wo, wm) / (4.0f * absdot(wo, wm)) * (R / (R + T));
    float dn = absdot(wi, wm) / sqr(dot(wi, wm) + dot(wo, wm) / etap);
    return mf.pdf(wo, wm) * dn * (T / (R + T));
}

// ---- ThinDielectricBxDF (dielectric.hpp:163-207): evaluate = {}, pdf = 0, sample = the specular pair with inter-reflection ----
JD bool thinDielectricSample(float eta, f3 wo, float uc, BSample &s) {
    float R = fresnelDielectric(wo.z, eta);
    float T = 1.0f - R;
    if (R < 1.0f) {
        R += (T * T * R) / (1.0f - R * R);
        T = 1.0f - R;
    }
    const float p = R / (R + T);
    if (uc < p) {
        f3 wi = mk3(-wo.x, -wo.y, wo.z);
        s.f = mk3(R / absCosTheta(wi)); s.wi = wi; s.pdf = p;
        return true;
    }
    f3 wi = -wo;
    s.f = mk3(T / absCosTheta(wi)); s.wi = wi; s.pdf = 1.0f - p;
    return true;
}

// ---- glTF metallic-roughness (gltf.hpp:9-123) ----
struct MR { GGX mf; f3 albedo; float metallic; };
JD f3 mrEval(const MR &b, f3 wo, f3 wi) {
    float co = absCosTheta(wo), ci = absCosTheta(wi);
    if (co == 0.0f || ci == 0.0f) return mk3(0.0f);
    f3 cDiff = lerp3(b.albedo, mk3(0.0f), b.metallic);
    f3 f0 = lerp3(mk3(0.04f), b.albedo, b.metallic);
    f3 wm = wi + wo;
    if (lenSqr(wm) == 0.0f) return mk3(0.0f);
    wm = normalize(wm);
    f3 F = schlick(wo, wm, f0);
    f3 fDiffuse = (1.0f - F) * cDiff * INV_PI;
    f3 fSpecular = b.mf.D(wm) * F * b.mf.G(wo, wi) / (4.0f * absCosTheta(wi) * absCosTheta(wo));
    return fDiffuse + fSpecular;
}
JD float mrSpecProb(const MR &b, f3 wo) {
    f3 f0 = lerp3(mk3(0.04f), b.albedo, b.metallic);
    f3 F = schlick(wo, mk3(0.0f, 0.0f, 1.0f), f0);
    float sw = (F.x + F.y + F.z) / 3.0f;
    float dw = (1.0f - b.metallic) * (1.0f - sw);
    float total = sw + dw;
    float p = 1.0f;
    if (total > 0.0f) p = sw / total;
    return p;
}
JD bool mrSample(const MR &b, f3 wo, float uc, f2 u, BSample &s) {
    float co = absCosTheta(wo);
    if (co == 0.0f) return false;
    f3 cDiff = lerp3(b.albedo, mk3(0.0f), b.metallic);
    f3 f0 = lerp3(mk3(0.04f), b.albedo, b.metallic);
    float p = mrSpecProb(b, wo);
    f3 wi, wm; float pdf;
    if (uc < p) {
        if (wo.z == 0.0f) return false;
        wm = b.mf.sampleWm(wo, u);
        wi = -wo + 2.0f * dot(wo, wm) * wm;
        if (!sameHemisphere(wo, wi)) return false;
        if (absCosTheta(wi) == 0.0f) return false;
        pdf = b.mf.pdf(wo, wm) / (4.0f * absdot(wo, wm));
    } else {
        wi = sampleCosineHemisphere(u);
        if (wo.z < 0.0f) wi.z *= -1.0f;
        wm = wi + wo;
        if (lenSqr(wm) == 0.0f) return false;
        wm = normalize(wm);
        pdf = cosineHemispherePDF(absCosTheta(wi));
    }
    f3 F = schlick(wo, wm, f0);
    f3 fDiffuse = (1.0f - F) * (cDiff / PI_F);
    f3 fSpecular = b.mf.D(wm) * F * b.mf.G(wo, wi) / (4.0f * absCosTheta(wi) * absCosTheta(wo));
    s.pdf = pdf; s.wi = wi; s.f = fDiffuse + fSpecular;
    return true;
}
JD float mrPdf(const MR &b, f3 wo, f3 wi) {
    if (!sameHemisphere(wo, wi)) return 0.0f;
    if (absCosTheta(wo) == 0.0f) return 0.0f;
    float p = mrSpecProb(b, wo);
    f3 wm = wi + wo;
    if (lenSqr(wm) == 0.0f) return 0.0f;
    wm = faceForward(normalize(wm), mk3(0.0f, 0.0f, 1.0f));
    float specularPdf = b.mf.pdf(wo, wm) / (4.0f * absdot(wo, wm));
    float diffusePdf = cosineHemispherePDF(absCosTheta(wi));
    return p * specularPdf + (1.0f - p) * diffusePdf;
}

// ---- textures (image.hpp:140-160): nearest, truncate toward zero, C-modulo wrap ----
JD f3 getTexel(const DTexture &t, const float *texels, bool linear, f2 uv) {
    int x = (int) (uv.x * (float) t.w);
    int y = (int) (uv.y * (float) t.h);
    int wu = x % t.w; if (wu < 0) wu += t.w;
    int wv = y % t.h; if (wv < 0) wv += t.h;
    const float *p = texels + (linear ? t.linear_off : t.raw_off) + (long long) (wv * t.w + wu) * t.c;
    return mk3(p[0], p[1], p[2]);
}

struct ShadeCtx {            // what the dispatch needs from the scene
    const DMaterial *materials;
    const DTexture  *textures;
    const float     *texels;
};

JD f3 a3(const float *p) { return mk3(p[0], p[1], p[2]); }

JD f3 albedoOf(const ShadeCtx &c, const DMaterial &m, f2 uv) {
    if (m.albedo_tex != -1) return getTexel(c.textures[m.albedo_tex], c.texels, true, uv);
    return a3(m.albedo);
}
JD void mrParams(const ShadeCtx &c, const DMaterial &m, f2 uv, float &metallic, float &roughness) {
    metallic = m.alpha_x; roughness = m.alpha_y;
    if (m.mr_tex != -1) { f3 mr = getTexel(c.textures[m.mr_tex], c.texels, false, uv); roughness = mr.y; metallic = mr.z; }
}

// MASK = bit set of Material::Type values the scene actually contains (scene_create knows them): the
// kernels are instantiated for "diffuse only" and "anything", so an all-Lambert scene such as the
// Cornell box does not carry the GGX / Fresnel code (registers, I-cache) it can never reach.
constexpr int MAT_ALL = 15, MAT_DIFFUSE_ONLY = 1, MAT_EVERY = 31;   // MAT_EVERY: + THIN_DIELECTRIC (the alternate-integrator kernels)

// sampleBxdf (bxdf.cpp:9-77)
template <int MASK = MAT_ALL>
JD bool sampleBxdf(const ShadeCtx &c, const DMaterial &m, f3 normal, f2 uv, f3 wo, float uc, f2 u, BSample &out) {
    Frame fr = Frame::fromZ(normal);
    f3 wol = fr.toLocal(wo);
    if (wol.z == 0.0f) return false;
    bool ok = false;
    out.specular = false;
    if ((MASK & 8) && m.type == 3) {
        float metallic, roughness; mrParams(c, m, uv, metallic, roughness);
        MR b; b.mf.ax = b.mf.ay = roughness * roughness; b.albedo = albedoOf(c, m, uv); b.metallic = metallic;
        ok = mrSample(b, wol, uc, u, out);
    } else if ((MASK & 1) && m.type == 0) {
        ok = diffuseSample(albedoOf(c, m, uv), wol, u, out);
    } else if ((MASK & 4) && m.type == 2) {
        GGX g; g.ax = m.alpha_x; g.ay = m.alpha_y;
        ok = conductorSample(g, a3(m.ior), a3(m.k), wol, u, out);
    } else if ((MASK & 2) && m.type == 1) {
        GGX g; g.ax = m.alpha_x; g.ay = m.alpha_y;
        ok = dielectricSample(g, m.ior[0], wol, uc, u, out);
    } else if ((MASK & 16) && m.type == 4) {
        ok = thinDielectricSample(m.ior[0], wol, uc, out);
    }
    if (!ok) return false;
    if (!nonzero(out.f) || out.pdf == 0.0f || out.wi.z == 0.0f) return false;
    out.wi = fr.toWorld(out.wi);
    return true;
}
// evalBxdf (bxdf.cpp:79-128) and pdfBxdf (bxdf.cpp:130-166) share the frame and the local vectors.
template <int MASK = MAT_ALL>
JD void evalPdfBxdf(const ShadeCtx &c, const DMaterial &m, f3 normal, f2 uv, f3 wo, f3 wi, f3 &f, float &pdf) {
    Frame fr = Frame::fromZ(normal);
    f3 wol = fr.toLocal(wo), wil = fr.toLocal(wi);
    f = mk3(0.0f); pdf = 0.0f;
    if (wol.z == 0.0f || wil.z == 0.0f) return;
    if ((MASK & 8) && m.type == 3) {
        float metallic, roughness; mrParams(c, m, uv, metallic, roughness);
        MR b; b.mf.ax = b.mf.ay = roughness * roughness; b.albedo = albedoOf(c, m, uv); b.metallic = metallic;
        f = mrEval(b, wol, wil);
        b.albedo = a3(m.albedo);                       // pdfBxdf uses the constant albedo (bxdf.cpp:146)
        pdf = mrPdf(b, wol, wil);
    } else if ((MASK & 1) && m.type == 0) {
        f = diffuseEval(albedoOf(c, m, uv), wol, wil);
        pdf = diffusePdf(wol, wil);
    } else if ((MASK & 4) && m.type == 2) {
        GGX g; g.ax = m.alpha_x; g.ay = m.alpha_y;
        f = conductorEval(g, a3(m.ior), a3(m.k), wol, wil);
        pdf = conductorPdf(g, wol, wil);
    } else if ((MASK & 2) && m.type == 1) {
        GGX g; g.ax = m.alpha_x; g.ay = m.alpha_y;
        f = dielectricEval(g, m.ior[0], wol, wil);
        pdf = dielectricPdf(g, m.ior[0], wol, wil);
    }
}

} // namespace jtx
