// micro-benchmark 6 (round 4): packed fp16 VALU (v_pk_fma_f16 / v_pk_max_f16 / v_pk_min_f16 / v_pk_add_f16), v_perm_b32, v_cvt_pkrtz_f16_f32 --
// the instructions a packed-fp16 wide-node test would be made of -- against the fp32 ones it would replace.  Same frame as rate4 / rate5:
// 8 waves per SIMD, 8 independent instructions per trip, shader clock read in the kernel.
// Also checks the semantics the design leans on: a zero-extended byte IS the fp16 subnormal q * 2^-24, v_pk_fma_f16 takes it at full
// precision (no flush), and v_perm_b32's selector convention.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define I8_2(S, TAIL) \
    asm volatile(S " %0, %8, %9 " TAIL "\n" S " %1, %9, %10 " TAIL "\n" S " %2, %10, %11 " TAIL "\n" S " %3, %11, %12 " TAIL "\n" \
                 S " %4, %12, %13 " TAIL "\n" S " %5, %13, %14 " TAIL "\n" S " %6, %14, %15 " TAIL "\n" S " %7, %15, %8 " TAIL \
        : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7))
#define I8_3(S, TAIL) \
    asm volatile(S " %0, %8, %9, %10 " TAIL "\n" S " %1, %9, %10, %11 " TAIL "\n" S " %2, %10, %11, %12 " TAIL "\n" S " %3, %11, %12, %13 " TAIL "\n" \
                 S " %4, %12, %13, %14 " TAIL "\n" S " %5, %13, %14, %15 " TAIL "\n" S " %6, %14, %15, %8 " TAIL "\n" S " %7, %15, %8, %9 " TAIL \
        : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7))
// the node-test shape: 8 different first operands, the SAME second and third operand (a, b of an axis)
#define I8_AB(S, TAIL) \
    asm volatile(S " %0, %8, %14, %15 " TAIL "\n" S " %1, %9, %14, %15 " TAIL "\n" S " %2, %10, %14, %15 " TAIL "\n" S " %3, %11, %14, %15 " TAIL "\n" \
                 S " %4, %12, %14, %15 " TAIL "\n" S " %5, %13, %14, %15 " TAIL "\n" S " %6, %8, %14, %15 " TAIL "\n" S " %7, %9, %14, %15 " TAIL \
        : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7))
// accumulating shape: w = op(w, r)   (max / min chains of the slab test run in place)
#define I8_ACC(S, TAIL) \
    asm volatile(S " %0, %0, %8 " TAIL "\n" S " %1, %1, %9 " TAIL "\n" S " %2, %2, %10 " TAIL "\n" S " %3, %3, %11 " TAIL "\n" \
                 S " %4, %4, %12 " TAIL "\n" S " %5, %5, %13 " TAIL "\n" S " %6, %6, %14 " TAIL "\n" S " %7, %7, %15 " TAIL \
        : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7))
template <int OP>
__global__ void __launch_bounds__(256, 8) k(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    // bit patterns that are harmless as fp16 pairs and as fp32: small positive numbers
    float r0 = __uint_as_float(0x3c003c00u + lane), r1 = __uint_as_float(0x3c013c01u + lane), r2 = __uint_as_float(0x38003800u + lane), r3 = __uint_as_float(0x34003400u + lane),
          r4 = __uint_as_float(0x3c003800u + lane), r5 = __uint_as_float(0x30003c00u + lane), r6 = __uint_as_float(0x3c003400u + lane), r7 = __uint_as_float(0x38003000u + lane);
    float w0 = r0, w1 = r1, w2 = r2, w3 = r3, w4 = r4, w5 = r5, w6 = r6, w7 = r7;
    const long long c_0 = clock64(), w_0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) I8_3("v_fma_f32", "");
        if (OP == 1) I8_3("v_pk_fma_f16", "");
        if (OP == 2) I8_2("v_pk_max_f16", "");
        if (OP == 3) I8_2("v_pk_min_f16", "");
        if (OP == 4) I8_2("v_pk_add_f16", "");
        if (OP == 5) I8_2("v_pk_add_f16", "neg_lo:[0,1] neg_hi:[0,1]");
        if (OP == 6) I8_3("v_perm_b32", "");
        if (OP == 7) I8_2("v_cvt_pkrtz_f16_f32", "");
        if (OP == 8) I8_AB("v_pk_fma_f16", "");
        if (OP == 9) I8_AB("v_fma_f32", "");
        if (OP == 10) I8_ACC("v_pk_max_f16", "");
        if (OP == 11) I8_ACC("v_max_f32", "");
        if (OP == 12) I8_2("v_pk_mul_f16", "");
        if (OP == 13) I8_2("v_and_b32", "");
        if (OP == 14) I8_3("v_bfe_u32", "");
        if (OP == 15) I8_3("v_and_or_b32", "");
        if (OP == 16) I8_AB("v_perm_b32", "");
        if (OP == 17) I8_3("v_med3_f32", "");
    }
    const long long c_1 = clock64(), w_1 = wall_clock64();
    if (blockIdx.x == 1000 && threadIdx.x == 0) { ((long long *) out)[0] = c_1 - c_0; ((long long *) out)[1] = w_1 - w_0; }
    out[1024 + blockIdx.x * 256 + threadIdx.x] = w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7;
}
// semantics
__global__ void ksem(const unsigned *in, unsigned *out) {
    const unsigned planes_lo = in[0], planes_hi = in[1], sel = in[2], a2 = in[3], b2 = in[4];
    unsigned p, f, mx, df;
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(p) : "v"(planes_hi), "v"(planes_lo), "v"(sel));
    asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(f) : "v"(p), "v"(a2), "v"(b2));
    asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(mx) : "v"(f), "v"(b2));
    asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(df) : "v"(f), "v"(b2));
    out[0] = p; out[1] = f; out[2] = mx; out[3] = df;
    float x = __uint_as_float(in[5]), y = __uint_as_float(in[6]); unsigned c;
    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(c) : "v"(x), "v"(y));
    out[4] = c;
}
static double g_ghz[32];
template <int OP> float run(float *d, int iters) {
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    k<OP><<<256 * 8, 256>>>(d, iters); (void) hipDeviceSynchronize();
    (void) hipEventRecord(e0); k<OP><<<256 * 8, 256>>>(d, iters); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; (void) hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    g_ghz[OP] = (double) h[0] / ((double) h[1] * 10.0);
    return ms;
}
static float h2f(unsigned short h) {        // fp16 -> float (host)
    const unsigned s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    float v;
    if (e == 0) v = (float) m * 5.9604644775390625e-08f;
    else if (e == 31) v = m ? __builtin_nanf("") : __builtin_inff();
    else { unsigned u = ((e + 112) << 23) | (m << 13); memcpy(&v, &u, 4); }
    return s ? -v : v;
}
int main() {
    float *d; (void) hipMalloc(&d, (1024 + 256 * 2048 * 8) * sizeof(float));
    const int it = 20000;
    const char *names[] = {"v_fma_f32 w,r,r',r''", "v_pk_fma_f16 w,r,r',r''", "v_pk_max_f16 w,r,r'", "v_pk_min_f16 w,r,r'", "v_pk_add_f16 w,r,r'", "v_pk_add_f16 w,r,-r'",
                           "v_perm_b32 w,r,r',r''", "v_cvt_pkrtz_f16_f32 w,r,r'", "v_pk_fma_f16 w,r[j],a,b", "v_fma_f32 w,r[j],a,b", "v_pk_max_f16 w,w,r", "v_max_f32 w,w,r",
                           "v_pk_mul_f16 w,r,r'", "v_and_b32 w,r,r'", "v_bfe_u32 w,r,r',r''", "v_and_or_b32 w,r,r',r''", "v_perm_b32 w,r[j],a,b", "v_med3_f32 w,r,r',r''"};
    float ms[18] = {run<0>(d, it), run<1>(d, it), run<2>(d, it), run<3>(d, it), run<4>(d, it), run<5>(d, it), run<6>(d, it), run<7>(d, it), run<8>(d, it), run<9>(d, it),
                    run<10>(d, it), run<11>(d, it), run<12>(d, it), run<13>(d, it), run<14>(d, it), run<15>(d, it), run<16>(d, it), run<17>(d, it)};
    for (int i = 0; i < 18; ++i) printf("%-30s %8.3f ms  %6.3f ns per wave-instruction per SIMD   shader clock %.2f GHz -> %.2f cycles\n", names[i], ms[i],
                                        ms[i] * 1e6 / (8.0 * 8 * it), g_ghz[i], ms[i] * 1e6 / (8.0 * 8 * it) * g_ghz[i]);
    // semantics: planes_lo holds children 0-3 = bytes 10, 20, 30, 40; planes_hi children 4-7 = 50, 60, 70, 255.
    // selector 0x0c070c02: lo half <- byte 2 (30), hi half <- byte 7 (255), the other bytes zero -> fp16 subnormals 30 * 2^-24 and 255 * 2^-24.
    // a = 2^14 in both halves (0x7000), b = 0.25 (0x3400): fma -> 30 * 2^-10 + 0.25 = 0.279296875, 255 * 2^-10 + 0.25 = 0.4990234375
    unsigned hin[8] = {0x281e140au, 0xff463c32u, 0x0c070c02u, 0x70007000u, 0x34003400u, 0, 0, 0};
    float fx = 1.00048828125f * 3.0f + 1e-4f, fy = -0.333333f; memcpy(&hin[5], &fx, 4); memcpy(&hin[6], &fy, 4);
    unsigned *din, *dout; (void) hipMalloc(&din, sizeof hin); (void) hipMalloc(&dout, 8 * sizeof(unsigned));
    (void) hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice);
    ksem<<<1, 1>>>(din, dout);
    unsigned ho[8]; (void) hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    printf("semantics: v_perm_b32(hi, lo, 0x0c070c02) = %08x (want 00ff001e)\n", ho[0]);
    printf("           v_pk_fma_f16(subnormal bytes, 2^14, 0.25) = %08x -> lo %.10g (want 0.279296875) hi %.10g (want 0.4990234375; fp16 rounds to 0.49902344)\n", ho[1], h2f(ho[1] & 0xffff), h2f(ho[1] >> 16));
    printf("           v_pk_max_f16(that, 0.25) = %08x ; v_pk_add_f16(that, -0.25) = %08x -> lo %.10g hi %.10g\n", ho[2], ho[3], h2f(ho[3] & 0xffff), h2f(ho[3] >> 16));
    printf("           v_cvt_pkrtz_f16_f32(%.9g, %.9g) = %08x -> lo %.10g hi %.10g (round toward zero)\n", fx, fy, ho[4], h2f(ho[4] & 0xffff), h2f(ho[4] >> 16));
    return 0;
}
