// micro-benchmark 7 (round 4): the instructions a sign-byte hit mask would be made of -- v_perm_b32 with the sign selectors (8-11),
// v_sad_u8, v_dot4_u32_u8, v_bitop3_b32, v_alignbit_b32, v_sub_f32 -- against what the node step uses today (v_cmp + v_cndmask + v_or3,
// v_bfe_u32).  Same frame as rate6: 8 waves per SIMD, 8 independent instructions per trip, shader clock read in the kernel.
// Semantics checked: which operand the sign selectors read, and that v_sad_u8 against zero sums the four bytes.
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
// w = op(r, r', w): accumulating third operand (v_sad_u8 / v_dot4 / v_bitop3 chains)
#define I8_ACC3(S, TAIL) \
    asm volatile(S " %0, %8, %9, %0 " TAIL "\n" S " %1, %9, %10, %1 " TAIL "\n" S " %2, %10, %11, %2 " TAIL "\n" S " %3, %11, %12, %3 " TAIL "\n" \
                 S " %4, %12, %13, %4 " TAIL "\n" S " %5, %13, %14, %5 " TAIL "\n" S " %6, %14, %15, %6 " TAIL "\n" S " %7, %15, %8, %7 " TAIL \
        : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7))
template <int OP>
__global__ void __launch_bounds__(256, 8) k(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    float r0 = __uint_as_float(0x3c003c00u + lane), r1 = __uint_as_float(0x3c013c01u + lane), r2 = __uint_as_float(0x38003800u + lane), r3 = __uint_as_float(0x34003400u + lane),
          r4 = __uint_as_float(0x3c003800u + lane), r5 = __uint_as_float(0x30003c00u + lane), r6 = __uint_as_float(0x3c003400u + lane), r7 = __uint_as_float(0x38003000u + lane);
    float w0 = r0, w1 = r1, w2 = r2, w3 = r3, w4 = r4, w5 = r5, w6 = r6, w7 = r7;
    const long long c_0 = clock64(), w_0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) I8_3("v_fma_f32", "");
        if (OP == 1) I8_2("v_sub_f32", "");
        if (OP == 2) I8_3("v_perm_b32", "");
        if (OP == 3) I8_3("v_sad_u8", "");
        if (OP == 4) I8_ACC3("v_sad_u8", "");
        if (OP == 5) I8_3("v_dot4_u32_u8", "");
        if (OP == 6) I8_3("v_bitop3_b32", "bitop3:0xe0");
        if (OP == 7) I8_3("v_alignbit_b32", "");
        if (OP == 8) I8_3("v_and_or_b32", "");
        if (OP == 9) I8_3("v_or3_b32", "");
        if (OP == 10) I8_3("v_lshl_or_b32", "");
        if (OP == 11) I8_3("v_max3_f32", "");
        if (OP == 12) I8_2("v_max_f32", "");
        if (OP == 13) I8_2("v_cvt_f32_ubyte1", "") ;
    }
    const long long c_1 = clock64(), w_1 = wall_clock64();
    if (blockIdx.x == 1000 && threadIdx.x == 0) { ((long long *) out)[0] = c_1 - c_0; ((long long *) out)[1] = w_1 - w_0; }
    out[1024 + blockIdx.x * 256 + threadIdx.x] = w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7;
}
template <> __global__ void __launch_bounds__(256, 8) k<13>(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    float r0 = __uint_as_float(0x3c003c00u + lane), r1 = __uint_as_float(0x3c013c01u + lane), r2 = __uint_as_float(0x38003800u + lane), r3 = __uint_as_float(0x34003400u + lane),
          r4 = __uint_as_float(0x3c003800u + lane), r5 = __uint_as_float(0x30003c00u + lane), r6 = __uint_as_float(0x3c003400u + lane), r7 = __uint_as_float(0x38003000u + lane);
    float w0 = r0, w1 = r1, w2 = r2, w3 = r3, w4 = r4, w5 = r5, w6 = r6, w7 = r7;
    const long long c_0 = clock64(), w_0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_cvt_f32_ubyte1 %0, %8\nv_cvt_f32_ubyte1 %1, %9\nv_cvt_f32_ubyte1 %2, %10\nv_cvt_f32_ubyte1 %3, %11\n"
                     "v_cvt_f32_ubyte1 %4, %12\nv_cvt_f32_ubyte1 %5, %13\nv_cvt_f32_ubyte1 %6, %14\nv_cvt_f32_ubyte1 %7, %15"
            : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7));
    }
    const long long c_1 = clock64(), w_1 = wall_clock64();
    if (blockIdx.x == 1000 && threadIdx.x == 0) { ((long long *) out)[0] = c_1 - c_0; ((long long *) out)[1] = w_1 - w_0; }
    out[1024 + blockIdx.x * 256 + threadIdx.x] = w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7;
}
// semantics
__global__ void ksem(const unsigned *in, unsigned *out) {
    const unsigned a = in[0], b = in[1];
    unsigned p0, p1, s, d, bo;
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(p0) : "v"(a), "v"(b), "v"(0x0b0a0908u));     // bytes: sel 8, 9, 10, 11
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(p1) : "v"(a), "v"(b), "v"(0x0c0c090bu));     // byte0 <- sel 11, byte1 <- sel 9, rest zero
    asm volatile("v_sad_u8 %0, %1, %2, %3" : "=v"(s) : "v"(in[2]), "v"(0u), "v"(in[3]));
    asm volatile("v_dot4_u32_u8 %0, %1, %2, %3" : "=v"(d) : "v"(in[2]), "v"(0x01010101u), "v"(in[3]));
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe0" : "=v"(bo) : "v"(in[4]), "v"(in[5]), "v"(in[6]));
    out[0] = p0; out[1] = p1; out[2] = s; out[3] = d; out[4] = bo;
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
    const char *names[] = {"v_fma_f32 w,r,r',r''", "v_sub_f32 w,r,r'", "v_perm_b32 w,r,r',r''", "v_sad_u8 w,r,r',r''", "v_sad_u8 w,r,r',w", "v_dot4_u32_u8 w,r,r',r''",
                           "v_bitop3_b32 w,r,r',r''", "v_alignbit_b32 w,r,r',r''", "v_and_or_b32 w,r,r',r''", "v_or3_b32 w,r,r',r''", "v_lshl_or_b32 w,r,r',r''",
                           "v_max3_f32 w,r,r',r''", "v_max_f32 w,r,r'", "v_cvt_f32_ubyte1 w,r"};
    float ms[14] = {run<0>(d, it), run<1>(d, it), run<2>(d, it), run<3>(d, it), run<4>(d, it), run<5>(d, it), run<6>(d, it), run<7>(d, it), run<8>(d, it), run<9>(d, it),
                    run<10>(d, it), run<11>(d, it), run<12>(d, it), run<13>(d, it)};
    for (int i = 0; i < 14; ++i) printf("%-30s %8.3f ms  %6.3f ns per wave-instruction per SIMD   shader clock %.2f GHz -> %.2f cycles\n", names[i], ms[i],
                                        ms[i] * 1e6 / (8.0 * 8 * it), g_ghz[i], ms[i] * 1e6 / (8.0 * 8 * it) * g_ghz[i]);
    // a = 0x80007fff (bit 31 set, bit 15 clear), b = 0x7fff8000 (bit 31 clear, bit 15 set)
    unsigned hin[8] = {0x80007fffu, 0x7fff8000u, 0x80402010u, 5u, 0x0000ff00u, 0x00ff0000u, 0x0f0f0f0fu, 0};
    unsigned *din, *dout; (void) hipMalloc(&din, sizeof hin); (void) hipMalloc(&dout, 8 * sizeof(unsigned));
    (void) hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice);
    ksem<<<1, 1>>>(din, dout);
    unsigned ho[8]; (void) hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    printf("semantics: v_perm_b32(a = 80007fff, b = 7fff8000, sel bytes 8,9,10,11) = %08x\n"
           "           (ISA text: 8 = sign of S1[15], 9 = S1[31], 10 = S0[15], 11 = S0[31]; S0 = a, S1 = b -> want byte0 ff, byte1 00, byte2 00, byte3 ff = ff0000ff)\n", ho[0]);
    printf("           v_perm_b32(a, b, 0x0c0c090b) = %08x (want 000000ff: byte0 <- sign(a) = ff, byte1 <- sign(b) = 00)\n", ho[1]);
    printf("           v_sad_u8(0x80402010, 0, 5) = %u (want 245 = 0x80 + 0x40 + 0x20 + 0x10 + 5)\n", ho[2]);
    printf("           v_dot4_u32_u8(0x80402010, 0x01010101, 5) = %u (want 245)\n", ho[3]);
    printf("           v_bitop3_b32(0000ff00, 00ff0000, 0f0f0f0f) bitop3:0xe0 = %08x\n", ho[4]);
    return 0;
}
