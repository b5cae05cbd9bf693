// jtx_device_math.hpp -- device-side fp32 vector math, RNG, warps and the shading frame for the
// gfx950 kernels.  Semantics = DESIGN.md "jtx math spec" (the reference's un-vendored jtx:: library,
// rt.hpp:3-21): componentwise, left to right, one IEEE rounding per written operation.  The file
// is compiled with -ffp-contract=off; never add fast-math intrinsics here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define JD __device__ __forceinline__

namespace jtx {

struct f3 { float x, y, z; };
struct f2 { float x, y; };

JD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
JD f3 mk3(float s) { return mk3(s, s, s); }
JD f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
JD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
JD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
JD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
JD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
JD f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
JD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
JD f3 operator-(float s, f3 a) { return mk3(s - a.x, s - a.y, s - a.z); }
JD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
JD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
JD float absdot(f3 a, f3 b) { return fabsf(dot(a, b)); }
JD f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
JD float lenSqr(f3 a) { return dot(a, a); }
JD float len(f3 a) { return sqrtf(lenSqr(a)); }
JD f3 normalize(f3 a) { return a / len(a); }
JD bool nonzero(f3 a) { return a.x != 0.0f || a.y != 0.0f || a.z != 0.0f; }   // Vec3::operator bool
JD float fmax2(float a, float b) { return a > b ? a : b; }
JD float fmin2(float a, float b) { return a < b ? a : b; }
JD float sqr(float x) { return x * x; }
JD float safeSqrt(float x) { return sqrtf(fmax2(0.0f, x)); }
JD float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
JD float lerpf(float a, float b, float t) { return (1.0f - t) * a + t * b; }
JD f3 lerp3(f3 a, f3 b, float t) { return (1.0f - t) * a + t * b; }
JD f3 faceForward(f3 n, f3 v) { return dot(n, v) < 0.0f ? -n : n; }
JD bool sameHemisphere(f3 a, f3 b) { return a.z * b.z > 0.0f; }
JD float absCosTheta(f3 w) { return fabsf(w.z); }
JD float cos2Theta(f3 w) { return w.z * w.z; }
JD float sin2Theta(f3 w) { return fmax2(0.0f, 1.0f - cos2Theta(w)); }
JD float sinTheta(f3 w) { return sqrtf(sin2Theta(w)); }
JD float tan2Theta(f3 w) { return sin2Theta(w) / cos2Theta(w); }
JD float cosPhi(f3 w) { float s = sinTheta(w); return s == 0.0f ? 1.0f : clampf(w.x / s, -1.0f, 1.0f); }
JD float sinPhi(f3 w) { float s = sinTheta(w); return s == 0.0f ? 0.0f : clampf(w.y / s, -1.0f, 1.0f); }
JD bool isinf_(float x) { return fabsf(x) == __builtin_inff(); }
// r + b*0 per component (what integrateMIS adds for an occluded light sample): r unless b is inf/NaN
JD f3 addBetaTimesZero(f3 r, f3 b) { return r + b * mk3(0.0f); }
JD int nonFiniteMask(f3 b) { return (fabsf(b.x) < __builtin_inff() ? 0 : 1) | (fabsf(b.y) < __builtin_inff() ? 0 : 2) | (fabsf(b.z) < __builtin_inff() ? 0 : 4); }
JD f3 poisonNonFinite(f3 r, int mask) {      // == r + b*0 given nonFiniteMask(b)
    const float qnan = __builtin_nanf("");
    return mk3((mask & 1) ? qnan : r.x, (mask & 2) ? qnan : r.y, (mask & 4) ? qnan : r.z);
}

constexpr float PI_F        = 3.14159265358979323846f;
constexpr float INV_PI      = 1.0f / PI_F;
constexpr float PI_OVER_4   = PI_F / 4;
constexpr float PI_OVER_2   = PI_F / 2;
constexpr float RAY_EPSILON = 1e-4f;          // scene.hpp:9

// Deterministic sin/cos, DESIGN.md "sincos spec".  k = floor(x*2/pi + 0.5); Cody-Waite in three
// steps; degree-7 / degree-8 polynomials on |r| <= pi/4.  Replaces jtx::sin / jtx::cos so that CPU
// oracle and GPU agree bit for bit (libm and ocml do not).
JD void det_sincos(float x, float &s, float &c) {
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float A = 1.5703125f;
    const float B = 4.837512969970703125e-4f;
    const float C = 7.54978995489188216e-8f;
    float q  = x * TWO_OVER_PI;
    float kf = floorf(q + 0.5f);
    int   k  = (int) kf;
    float r  = ((x - kf * A) - kf * B) - kf * C;
    float z  = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    int m = k & 3;
    float ss = (m & 1) ? cp : sp;
    float cc = (m & 1) ? sp : cp;
    s = (m & 2) ? -ss : ss;
    c = (m == 1 || m == 2) ? -cc : cc;
}

// PCG RXS-M-XS-32 with the reference's ">> 2" output shift (util/rand.hpp:99-104)
JD uint32_t fnv1a_3(uint32_t x, uint32_t y, uint32_t n) {
    uint32_t h = 2166136261u;
    h ^= x; h *= 16777619u;
    h ^= y; h *= 16777619u;
    h ^= n; h *= 16777619u;
    return h;
}
struct Rng {
    uint32_t state;
    JD void seed(uint32_t x, uint32_t y, uint32_t n) {     // RNG(x,y,n) with state_ read as 0 (quirk Q1)
        state = 0;
        advance();
        state += fnv1a_3(x, y, n);
        advance();
    }
    JD uint32_t advance() {
        uint32_t s = state;
        state = state * 747796405u + 2891336453u;
        uint32_t word = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
        return (word >> 2u) ^ word;
    }
    JD float f() { return (float) (advance() & 0xFFFFFFu) / 16777216.0f; }
    JD uint32_t sampleRange(int range) {                   // one advance; hi32(x*range) (quirk Q2)
        uint32_t x = advance();
        if (range <= 0) return 0u;
        return __umulhi(x, (uint32_t) range);
    }
};

JD f2 sampleUniformDiskPolar(f2 u) {                        // sampling.hpp:22-26
    float r = sqrtf(u.x);
    float theta = 2.0f * PI_F * u.y;
    float s, c; det_sincos(theta, s, c);
    return mk2(r * c, r * s);
}
JD f2 sampleUniformDiskConcentric(f2 u) {                   // sampling.hpp:28-46
    float ox = 2.0f * u.x - 1.0f, oy = 2.0f * u.y - 1.0f;
    if (ox == 0.0f && oy == 0.0f) return mk2(0.0f, 0.0f);
    float r, theta;
    if (fabsf(ox) > fabsf(oy)) { r = ox; theta = PI_OVER_4 * (oy / ox); }
    else                       { r = oy; theta = PI_OVER_2 - PI_OVER_4 * (ox / oy); }
    float s, c; det_sincos(theta, s, c);
    return mk2(r * c, r * s);
}
JD f3 sampleCosineHemisphere(f2 u) {                        // sampling.hpp:56-59
    f2 d = sampleUniformDiskConcentric(u);
    return mk3(d.x, d.y, safeSqrt(1.0f - d.x * d.x - d.y * d.y));
}
JD float cosineHemispherePDF(float c) { return c * INV_PI; }

// Frame::fromZ: Duff et al. branchless ONB (bxdf.cpp:10,80,131)
struct Frame {
    f3 x, y, z;
    JD static Frame fromZ(f3 n) {
        float sign = copysignf(1.0f, n.z);
        float a = -1.0f / (sign + n.z);
        float b = n.x * n.y * a;
        Frame f;
        f.x = mk3(1.0f + sign * sqr(n.x) * a, sign * b, -sign * n.x);
        f.y = mk3(b, sign + sqr(n.y) * a, -n.y);
        f.z = n;
        return f;
    }
    JD f3 toLocal(f3 v) const { return mk3(dot(v, x), dot(v, y), dot(v, z)); }
    JD f3 toWorld(f3 v) const { return v.x * x + v.y * y + v.z * z; }
};

} // namespace jtx
