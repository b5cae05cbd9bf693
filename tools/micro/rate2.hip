// micro-benchmark 2: operand forms (gfx950).  8 chains per lane, 8 waves per SIMD, every CU busy; ns per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void __launch_bounds__(256, 8) k(float *out, int iters, float c0, float c1, unsigned long long m) {
    const int lane = threadIdx.x & 63;
    float a[8], b[8]; unsigned u[8];
    for (int j = 0; j < 8; ++j) { a[j] = lane + j; b[j] = lane * 3 + j; u[j] = lane * 77u + j; }
    unsigned long long sm = m;      // a lane mask in an SGPR pair
    asm volatile("s_mov_b64 %0, %1" : "=s"(sm) : "s"(m));
    float sc = c0; asm volatile("s_mov_b32 %0, %1" : "=s"(sc) : "s"(c0));
    const long long c_0 = clock64(), w_0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c0), "v"(c1));
            if (OP == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c0), "s"(sm));
            if (OP == 2) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(c0));
            if (OP == 3) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b[j]));
            if (OP == 4) asm volatile("v_min_f32 %0, %0, %0" : "+v"(a[j]));
            if (OP == 5) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[j]) : "s"(sc));
            if (OP == 6) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[j]) : "s"(sc));
            if (OP == 7) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[j]) : "s"(sc));
            if (OP == 8) asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(sm) : "v"(a[j]), "v"(c0));
            if (OP == 9) asm volatile("v_cmp_le_f32_e32 vcc, %0, %1\n v_cndmask_b32_e64 %2, 0, 1, vcc" : : "v"(a[j]), "v"(c0), "v"(u[j]) : "vcc");
            if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b[j]), "v"(c1));
            if (OP == 11) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c0));
            if (OP == 12) asm volatile("v_max_f32 %0, 0x3a83126f, %0" : "+v"(a[j]));
            if (OP == 13) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(u[(j + 1) & 7]), "v"(u[(j + 2) & 7]));
            if (OP == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(a[j]) : "v"(b[j]));
            if (OP == 15) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[j]) : "v"(c0), "v"(c1));
        }
    }
    const long long c_1 = clock64(), w_1 = wall_clock64();
    if (blockIdx.x == 1000 && threadIdx.x == 0) { ((long long *) out)[0] = c_1 - c_0; ((long long *) out)[1] = w_1 - w_0; }
    float s = 0; for (int j = 0; j < 8; ++j) s += a[j] + b[j] + (float) u[j];
    out[1024 + blockIdx.x * 256 + threadIdx.x] = s + (float) (sm & 1);
}
static double g_ghz[16];
template <int OP> float run(float *d, int iters) {
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    k<OP><<<256 * 8, 256>>>(d, iters, 1.0001f, 0.5f, 0x5555555555555555ull); (void) hipDeviceSynchronize();
    (void) hipEventRecord(e0); k<OP><<<256 * 8, 256>>>(d, iters, 1.0001f, 0.5f, 0x5555555555555555ull); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; (void) hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    g_ghz[OP] = (double) h[0] / ((double) h[1] * 10.0);      // shader clocks per 10-ns tick of the 100 MHz wall clock
    return ms;
}
int main() {
    float *d; (void) hipMalloc(&d, (1024 + 256 * 2048 * 8) * sizeof(float));
    const int it = 20000;
    const char *names[] = {"v_fma v,v(c),v(c)", "v_cndmask e64 sgpr-pair", "v_cndmask e32 vcc", "v_min v,v", "v_min a,a,a", "v_min sgpr,v", "v_mul sgpr,v", "v_sub sgpr,v",
                           "v_cmp e64 -> sgpr", "v_cmp vcc + cndmask (2 instr)", "v_fma v,v,v", "v_add v,v", "v_max literal,v", "v_or3", "v_mov", "v_fmac"};
    float ms[16] = {run<0>(d, it), run<1>(d, it), run<2>(d, it), run<3>(d, it), run<4>(d, it), run<5>(d, it), run<6>(d, it), run<7>(d, it), run<8>(d, it),
                    run<9>(d, it), run<10>(d, it), run<11>(d, it), run<12>(d, it), run<13>(d, it), run<14>(d, it), run<15>(d, it)};
    for (int i = 0; i < 16; ++i) printf("%-32s %8.3f ms  %6.3f ns per wave-instruction per SIMD   shader clock %.2f GHz -> %.2f cycles\n", names[i], ms[i], ms[i] * 1e6 / (8.0 * 8 * it) / (i == 9 ? 2 : 1), g_ghz[i], ms[i] * 1e6 / (8.0 * 8 * it) / (i == 9 ? 2 : 1) * g_ghz[i]);
    return 0;
}
