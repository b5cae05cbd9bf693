// micro-benchmark 4: no dependent chains at all -- every instruction reads a read-only register pool r[0..7] and writes its own
// destination w[j]; K = distance (in registers, i.e. VGPR banks mod 4) between the two sources.
#include <hip/hip_runtime.h>
#include <cstdio>
#define I8(K, S) \
    asm volatile(S " %0, %8, %9\n" S " %1, %9, %10\n" S " %2, %10, %11\n" S " %3, %11, %12\n" S " %4, %12, %13\n" S " %5, %13, %14\n" S " %6, %14, %15\n" S " %7, %15, %8" \
        : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7))
#define I8S(S) \
    asm volatile(S " %0, %8, %8\n" S " %1, %9, %9\n" S " %2, %10, %10\n" S " %3, %11, %11\n" S " %4, %12, %12\n" S " %5, %13, %13\n" S " %6, %14, %14\n" S " %7, %15, %15" \
        : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7))
#define I8_3(S) \
    asm volatile(S " %0, %8, %9, %10\n" S " %1, %9, %10, %11\n" S " %2, %10, %11, %12\n" S " %3, %11, %12, %13\n" S " %4, %12, %13, %14\n" S " %5, %13, %14, %15\n" S " %6, %14, %15, %8\n" S " %7, %15, %8, %9" \
        : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7))
#define I8_K(S, KS) \
    asm volatile(S " %0, " KS ", %8\n" S " %1, " KS ", %9\n" S " %2, " KS ", %10\n" S " %3, " KS ", %11\n" S " %4, " KS ", %12\n" S " %5, " KS ", %13\n" S " %6, " KS ", %14\n" S " %7, " KS ", %15" \
        : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7))
#define I8_1(S) \
    asm volatile(S " %0, %8\n" S " %1, %9\n" S " %2, %10\n" S " %3, %11\n" S " %4, %12\n" S " %5, %13\n" S " %6, %14\n" S " %7, %15" \
        : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7))
template <int OP>
__global__ void __launch_bounds__(256, 8) k(float *out, int iters, float c0) {
    const int lane = threadIdx.x & 63;
    float r0 = lane, r1 = lane + 1, r2 = lane + 2, r3 = lane + 3, r4 = lane + 4, r5 = lane + 5, r6 = lane + 6, r7 = lane + 7;
    float w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, w6 = 0, w7 = 0;
    const long long c_0 = clock64(), w_0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) I8(1, "v_add_f32");
        if (OP == 1) I8(1, "v_mul_f32");
        if (OP == 2) I8(1, "v_max_f32");
        if (OP == 3) I8(1, "v_min_f32");
        if (OP == 4) I8S("v_max_f32");
        if (OP == 5) I8_3("v_fma_f32");
        if (OP == 6) I8_3("v_max3_f32");
        if (OP == 7) I8_1("v_cvt_f32_ubyte1");
        if (OP == 8) I8_K("v_sub_f32", "s4");
        if (OP == 9) I8_K("v_max_f32", "0x3a83126f");
        if (OP == 10) I8_K("v_add_f32", "1.0");
        if (OP == 11) I8_1("v_mov_b32");
        if (OP == 12) I8(1, "v_and_b32");
        if (OP == 13) I8_1("v_rcp_f32");
    }
    const long long c_1 = clock64(), w_1 = wall_clock64();
    if (blockIdx.x == 1000 && threadIdx.x == 0) { ((long long *) out)[0] = c_1 - c_0; ((long long *) out)[1] = w_1 - w_0; }
    out[1024 + blockIdx.x * 256 + threadIdx.x] = w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7;
}
static double g_ghz[16];
template <int OP> float run(float *d, int iters) {
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    k<OP><<<256 * 8, 256>>>(d, iters, 1.0001f); (void) hipDeviceSynchronize();
    (void) hipEventRecord(e0); k<OP><<<256 * 8, 256>>>(d, iters, 1.0001f); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; (void) hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    g_ghz[OP] = (double) h[0] / ((double) h[1] * 10.0);
    return ms;
}
int main() {
    float *d; (void) hipMalloc(&d, (1024 + 256 * 2048 * 8) * sizeof(float));
    const int it = 20000;
    const char *names[] = {"v_add w,r,r'", "v_mul w,r,r'", "v_max w,r,r'", "v_min w,r,r'", "v_max w,r,r", "v_fma w,r,r',r''", "v_max3 w,r,r',r''", "v_cvt_f32_ubyte1 w,r",
                           "v_sub w,sgpr,r", "v_max w,literal,r", "v_add w,1.0,r", "v_mov w,r", "v_and w,r,r'", "v_rcp w,r"};
    float ms[14] = {run<0>(d, it), run<1>(d, it), run<2>(d, it), run<3>(d, it), run<4>(d, it), run<5>(d, it), run<6>(d, it), run<7>(d, it), run<8>(d, it),
                    run<9>(d, it), run<10>(d, it), run<11>(d, it), run<12>(d, it), run<13>(d, it)};
    for (int i = 0; i < 14; ++i) printf("%-24s %8.3f ms  %6.3f ns per wave-instruction per SIMD   shader clock %.2f GHz -> %.2f cycles\n", names[i], ms[i],
                                        ms[i] * 1e6 / (8.0 * 8 * it), g_ghz[i], ms[i] * 1e6 / (8.0 * 8 * it) * g_ghz[i]);
    return 0;
}
