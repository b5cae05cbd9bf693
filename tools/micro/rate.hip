// micro-benchmark: issue cost of the VALU instructions the traversal kernels are made of (gfx950), 8 independent chains per
// lane, 8 waves per SIMD, every CU busy.  Prints ns per wave-instruction per SIMD relative to v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void __launch_bounds__(256, 8) k(float *out, int iters, float c0, float c1) {
    const int lane = threadIdx.x & 63;
    float a[8]; v2f p[8]; unsigned u[8];
    for (int j = 0; j < 8; ++j) { a[j] = lane + j; p[j].x = lane + j; p[j].y = lane - j; u[j] = lane * 77u + j; }
    v2f cc; cc.x = c0; cc.y = c1;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c0), "v"(c1));
            if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c0));
            if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(cc));
            if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j]) : "v"(cc));
            if (OP == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c0), "v"(c1));
            if (OP == 5) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[j]) : "v"(u[j]));
            if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(c0));
            if (OP == 7) asm volatile("v_bfe_u32 %0, %0, 3, 3" : "+v"(u[j]));
            if (OP == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
            if (OP == 9) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[j]));
            if (OP == 10) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j]) : "v"(cc));
            if (OP == 11) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
            if (OP == 12) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
            if (OP == 13) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c0));
            if (OP == 14) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[j]), "v"(c0) : "vcc");
            if (OP == 15) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 7]));
        }
    }
    float s = 0; for (int j = 0; j < 8; ++j) s += a[j] + p[j].x + p[j].y + (float) u[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> float run(float *d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256 * 8, 256>>>(d, iters, 1.0001f, 0.5f); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<256 * 8, 256>>>(d, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float *d; hipMalloc(&d, 256 * 2048 * 8 * sizeof(float));
    const int it = 40000;
    const char *names[] = {"v_fma_f32", "v_mul_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_max3_f32", "v_cvt_f32_ubyte1", "v_cndmask_b32", "v_bfe_u32", "v_mul_lo_u32",
                           "v_rcp_f32", "v_pk_add_f32", "v_dot4_u32_u8", "v_perm_b32", "v_min_f32", "v_cmp_le_f32", "v_lshl_or_b32"};
    float ms[16] = {run<0>(d, it), run<1>(d, it), run<2>(d, it), run<3>(d, it), run<4>(d, it), run<5>(d, it), run<6>(d, it), run<7>(d, it), run<8>(d, it),
                    run<9>(d, it), run<10>(d, it), run<11>(d, it), run<12>(d, it), run<13>(d, it), run<14>(d, it), run<15>(d, it)};
    // 8 waves x 8 workgroup-waves... per SIMD: 2048 workgroups x 4 waves / (256 CUs x 4 SIMDs) = 8 waves per SIMD, each 8 * iters instructions
    printf("v_fma_f32 again    %8.3f ms\n", run<0>(d, it));
    for (int i = 0; i < 16; ++i) printf("%-18s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (2.4 GHz)  x%.2f of v_fma_f32\n", names[i], ms[i],
                                        ms[i] * 1e-3 * 2.4e9 / (8.0 * 8 * it), ms[i] / ms[0]);
    return 0;
}
