// micro-benchmark 8 (round 5): can the 48 byte -> float conversions of an 8-ary node step (v_cvt_f32_ubyteK, 4.4 cycles) be had at the
// 2.4-cycle rate of v_mul_f32?  A zero-extended byte q read as an fp32 is the denormal q * 2^-149; v_mul_f32 through SDWA takes the byte
// straight out of the packed word and q * 2^-149 * 2^127 = q * 2^-22 is exact, so fma(q * 2^-22, a * 2^22, b) rounds like fma(q, a, b).
// Also: what a correctly rounded division's v_div_scale / v_div_fmas / v_div_fixup cost, and v_addc_co_u32 (the leaf list's mask).
// Same frame as rate6 / rate7: 8 waves per SIMD, 8 independent instructions per trip, shader clock read in the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define I8_1(S, TAIL) \
    asm volatile(S " %0, %8 " TAIL "\n" S " %1, %9 " TAIL "\n" S " %2, %10 " TAIL "\n" S " %3, %11 " TAIL "\n" \
                 S " %4, %12 " TAIL "\n" S " %5, %13 " TAIL "\n" S " %6, %14 " TAIL "\n" S " %7, %15 " TAIL \
        : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7))
#define I8_2(S, TAIL) \
    asm volatile(S " %0, %8, %9 " TAIL "\n" S " %1, %9, %10 " TAIL "\n" S " %2, %10, %11 " TAIL "\n" S " %3, %11, %12 " TAIL "\n" \
                 S " %4, %12, %13 " TAIL "\n" S " %5, %13, %14 " TAIL "\n" S " %6, %14, %15 " TAIL "\n" S " %7, %15, %8 " TAIL \
        : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7))
#define I8_3(S, TAIL) \
    asm volatile(S " %0, %8, %9, %10 " TAIL "\n" S " %1, %9, %10, %11 " TAIL "\n" S " %2, %10, %11, %12 " TAIL "\n" S " %3, %11, %12, %13 " TAIL "\n" \
                 S " %4, %12, %13, %14 " TAIL "\n" S " %5, %13, %14, %15 " TAIL "\n" S " %6, %14, %15, %8 " TAIL "\n" S " %7, %15, %8, %9 " TAIL \
        : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7))
// v_div_scale writes an SGPR pair too
#define I8_DS \
    asm volatile("v_div_scale_f32 %0, vcc, %8, %8, %9\nv_div_scale_f32 %1, vcc, %9, %9, %10\nv_div_scale_f32 %2, vcc, %10, %10, %11\nv_div_scale_f32 %3, vcc, %11, %11, %12\n" \
                 "v_div_scale_f32 %4, vcc, %12, %12, %13\nv_div_scale_f32 %5, vcc, %13, %13, %14\nv_div_scale_f32 %6, vcc, %14, %14, %15\nv_div_scale_f32 %7, vcc, %15, %15, %8" \
        : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7) : "vcc")
#define I8_ADDC \
    asm volatile("v_addc_co_u32 %0, vcc, %8, %9, vcc\nv_addc_co_u32 %1, vcc, %9, %10, vcc\nv_addc_co_u32 %2, vcc, %10, %11, vcc\nv_addc_co_u32 %3, vcc, %11, %12, vcc\n" \
                 "v_addc_co_u32 %4, vcc, %12, %13, vcc\nv_addc_co_u32 %5, vcc, %13, %14, vcc\nv_addc_co_u32 %6, vcc, %14, %15, vcc\nv_addc_co_u32 %7, vcc, %15, %8, vcc" \
        : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7) : "vcc")
template <int OP>
__global__ void __launch_bounds__(256, 8) k(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    // bytes of every operand are small integers (denormals when read as floats); for the float ops use ordinary values
    const bool bytes = OP == 1 || OP == 2 || OP == 3 || OP == 4 || OP == 9;
    float r0, r1, r2, r3, r4, r5, r6, r7;
    if (bytes) { r0 = __uint_as_float(0x11223344u + lane); r1 = __uint_as_float(0x01020304u + lane); r2 = __uint_as_float(0x7f3f1f0fu); r3 = __uint_as_float(0x10203040u + lane);
                 r4 = __uint_as_float(0x55667788u); r5 = __uint_as_float(0x0a0b0c0du + lane); r6 = __uint_as_float(0x60504030u); r7 = __uint_as_float(0x21436587u); }
    else { r0 = 1.0f + lane; r1 = 1.5f + lane; r2 = 0.5f + lane; r3 = 0.25f; r4 = 3.0f; r5 = 0.125f + lane; r6 = 7.0f; r7 = 0.75f; }
    float w0 = r0, w1 = r1, w2 = r2, w3 = r3, w4 = r4, w5 = r5, w6 = r6, w7 = r7;
    const float c127 = __uint_as_float(254u << 23);      // 2^127
    const long long c_0 = clock64(), w_0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) I8_2("v_mul_f32", "");
        if (OP == 1) I8_1("v_cvt_f32_ubyte1", "");
        if (OP == 2) {   // w = c127 * byte1(r)
            asm volatile("v_mul_f32_sdwa %0, %16, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         "v_mul_f32_sdwa %1, %16, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         "v_mul_f32_sdwa %2, %16, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         "v_mul_f32_sdwa %3, %16, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         "v_mul_f32_sdwa %4, %16, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         "v_mul_f32_sdwa %5, %16, %13 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         "v_mul_f32_sdwa %6, %16, %14 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                         "v_mul_f32_sdwa %7, %16, %15 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"
                : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "v"(c127));
        }
        if (OP == 3) I8_1("v_cvt_f32_u32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1");
        if (OP == 4) I8_3("v_fma_f32", "");                 // fma on denormal operands (all three are byte words)
        if (OP == 5) I8_DS;
        if (OP == 6) I8_3("v_div_fmas_f32", "");
        if (OP == 7) I8_3("v_div_fixup_f32", "");
        if (OP == 8) I8_ADDC;
        if (OP == 9) I8_2("v_mul_f32", "");                 // plain v_mul_f32 on denormal operands
        if (OP == 10) I8_2("v_ldexp_f32", "");
        if (OP == 11) I8_1("v_rcp_f32", "");
        if (OP == 12) I8_1("v_sqrt_f32", "");
        if (OP == 13) I8_2("v_add_u32", "");
        if (OP == 14) I8_2("v_lshlrev_b32", "");
        if (OP == 15) I8_3("v_bfe_u32", "");
    }
    const long long c_1 = clock64(), w_1 = wall_clock64();
    if (blockIdx.x == 1000 && threadIdx.x == 0) { ((long long *) out)[0] = c_1 - c_0; ((long long *) out)[1] = w_1 - w_0; }
    out[1024 + blockIdx.x * 256 + threadIdx.x] = w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7;
}
// semantics: c * byte_k(word) through SDWA against (float) byte * 2^-22, and the fma built on it against fma((float) q, a, b)
__global__ void ksem(const unsigned *in, float *out) {
    const unsigned word = in[0];
    const float c127 = __uint_as_float(254u << 23), a = __uint_as_float(in[1]), b = __uint_as_float(in[2]);
    float m0, m1, m2, m3;
    asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(m0) : "v"(c127), "v"(word));
    asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(m1) : "v"(c127), "v"(word));
    asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(m2) : "v"(c127), "v"(word));
    asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(m3) : "v"(c127), "v"(word));
    out[0] = m0; out[1] = m1; out[2] = m2; out[3] = m3;
    const float a22 = a * 4194304.0f;
    out[4] = __fmaf_rn(m1, a22, b);
    out[5] = __fmaf_rn((float) ((word >> 8) & 0xffu), a, b);
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
int main() {
    float *d; (void) hipMalloc(&d, (1024 + 256 * 2048 * 8) * sizeof(float));
    const int it = 20000;
    const char *names[] = {"v_mul_f32 w,r,r'", "v_cvt_f32_ubyte1 w,r", "v_mul_f32_sdwa w,2^127,r.byte1", "v_cvt_f32_u32_sdwa w,r.byte1", "v_fma_f32 (denormal operands)",
                           "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_addc_co_u32", "v_mul_f32 (denormal operands)", "v_ldexp_f32", "v_rcp_f32", "v_sqrt_f32",
                           "v_add_u32", "v_lshlrev_b32", "v_bfe_u32"};
    float ms[16] = {run<0>(d, it), run<1>(d, it), run<2>(d, it), run<3>(d, it), run<4>(d, it), run<5>(d, it), run<6>(d, it), run<7>(d, it), run<8>(d, it), run<9>(d, it),
                    run<10>(d, it), run<11>(d, it), run<12>(d, it), run<13>(d, it), run<14>(d, it), run<15>(d, it)};
    for (int i = 0; i < 16; ++i) printf("%-32s %8.3f ms  %6.3f ns per wave-instruction per SIMD   shader clock %.2f GHz -> %.2f cycles\n", names[i], ms[i],
                                        ms[i] * 1e6 / (8.0 * 8 * it), g_ghz[i], ms[i] * 1e6 / (8.0 * 8 * it) * g_ghz[i]);
    unsigned hin[4] = {0xc8ff0107u, 0, 0, 0};
    const float a = 0.0123456789f, b = -3.14159274f; memcpy(&hin[1], &a, 4); memcpy(&hin[2], &b, 4);
    unsigned *din; float *dout; (void) hipMalloc(&din, sizeof hin); (void) hipMalloc(&dout, 8 * sizeof(float));
    (void) hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice);
    ksem<<<1, 1>>>(din, dout);
    float ho[8]; (void) hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    printf("semantics: 2^127 * bytes of c8ff0107 through SDWA, times 2^22: %g %g %g %g (want 7 1 255 200)\n", ho[0] * 4194304.0f, ho[1] * 4194304.0f, ho[2] * 4194304.0f, ho[3] * 4194304.0f);
    unsigned u4, u5; memcpy(&u4, &ho[4], 4); memcpy(&u5, &ho[5], 4);
    printf("           fma(byte1 * 2^-22, a * 2^22, b) = %08x, fma((float) byte1, a, b) = %08x (%s)\n", u4, u5, u4 == u5 ? "equal" : "DIFFERENT");
    return 0;
}
