// micro-benchmark 5: v_fma_mix_f32 (an f16 operand converted inside the fma) against v_cvt_f32_ubyteN + v_fma_f32, and v_cvt_f32_f16.
#include <hip/hip_runtime.h>
#include <cstdio>
#define I8(S, TAIL) \
    asm volatile(S " %0, %8, %9, %10 " TAIL "\n" S " %1, %9, %10, %11 " TAIL "\n" S " %2, %10, %11, %12 " TAIL "\n" S " %3, %11, %12, %13 " TAIL "\n" \
                 S " %4, %12, %13, %14 " TAIL "\n" S " %5, %13, %14, %15 " TAIL "\n" S " %6, %14, %15, %8 " TAIL "\n" S " %7, %15, %8, %9 " TAIL \
        : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7))
#define I8_1(S) \
    asm volatile(S " %0, %8\n" S " %1, %9\n" S " %2, %10\n" S " %3, %11\n" S " %4, %12\n" S " %5, %13\n" S " %6, %14\n" S " %7, %15" \
        : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7))
template <int OP>
__global__ void __launch_bounds__(256, 8) k(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    float r0 = lane, r1 = lane + 1, r2 = lane + 2, r3 = lane + 3, r4 = lane + 4, r5 = lane + 5, r6 = lane + 6, r7 = lane + 7;
    float w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, w6 = 0, w7 = 0;
    const long long c_0 = clock64(), w_0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) I8("v_fma_f32", "");
        if (OP == 1) I8("v_fma_mix_f32", "op_sel:[0,0,0] op_sel_hi:[1,0,0]");
        if (OP == 2) I8("v_fma_mix_f32", "op_sel:[1,0,0] op_sel_hi:[1,0,0]");
        if (OP == 3) I8_1("v_cvt_f32_f16");
        if (OP == 4) I8_1("v_cvt_f32_ubyte2");
        if (OP == 5) I8("v_fma_mix_f32", "op_sel:[0,0,0] op_sel_hi:[0,0,0]");
    }
    const long long c_1 = clock64(), w_1 = wall_clock64();
    if (blockIdx.x == 1000 && threadIdx.x == 0) { ((long long *) out)[0] = c_1 - c_0; ((long long *) out)[1] = w_1 - w_0; }
    out[1024 + blockIdx.x * 256 + threadIdx.x] = w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7;
}
// semantics: fma_mix with src0 = f16 half
__global__ void ksem(const unsigned *in, float a, float b, float *out) {
    unsigned x = in[threadIdx.x]; float lo, hi;
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(x), "v"(a), "v"(b));
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(x), "v"(a), "v"(b));
    out[2 * threadIdx.x] = lo; out[2 * threadIdx.x + 1] = hi;
}
static double g_ghz[8];
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
    const char *names[] = {"v_fma_f32", "v_fma_mix_f32 src0 = f16 lo", "v_fma_mix_f32 src0 = f16 hi", "v_cvt_f32_f16", "v_cvt_f32_ubyte2", "v_fma_mix_f32 all f32"};
    float ms[6] = {run<0>(d, it), run<1>(d, it), run<2>(d, it), run<3>(d, it), run<4>(d, it), run<5>(d, it)};
    for (int i = 0; i < 6; ++i) printf("%-30s %8.3f ms  %6.3f ns per wave-instruction per SIMD   shader clock %.2f GHz -> %.2f cycles\n", names[i], ms[i],
                                       ms[i] * 1e6 / (8.0 * 8 * it), g_ghz[i], ms[i] * 1e6 / (8.0 * 8 * it) * g_ghz[i]);
    // semantics: halves 0x6400 (1024.0) / 0x4500 (5.0): fma(h, 2, 0.5)
    unsigned hin[64]; for (int i = 0; i < 64; ++i) hin[i] = 0x64004500u + (unsigned) i + ((unsigned) i << 16);
    unsigned *din; float *dout; (void) hipMalloc(&din, sizeof hin); (void) hipMalloc(&dout, 128 * sizeof(float));
    (void) hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice);
    ksem<<<1, 64>>>(din, 2.0f, 0.5f, dout);
    float ho[128]; (void) hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    printf("semantics: lo half 0x4500 (5.0) -> %g (want 10.5), hi half 0x6400 (1024.0) -> %g (want 2048.5); lane 3: %g %g\n", ho[0], ho[1], ho[6], ho[7]);
    return 0;
}
