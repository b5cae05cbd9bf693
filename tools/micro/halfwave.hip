// micro-benchmark: does a wave64 VALU instruction cost less when one 32-lane half of EXEC is empty?  (gfx950, SIMD-32)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long mask, int iters) {
    const int lane = threadIdx.x & 63;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
            a0 = __fmaf_rn(a0, 1.0001f, 0.5f); a1 = __fmaf_rn(a1, 1.0001f, 0.5f); a2 = __fmaf_rn(a2, 1.0001f, 0.5f); a3 = __fmaf_rn(a3, 1.0001f, 0.5f);
            a4 = __fmaf_rn(a4, 1.0001f, 0.5f); a5 = __fmaf_rn(a5, 1.0001f, 0.5f); a6 = __fmaf_rn(a6, 1.0001f, 0.5f); a7 = __fmaf_rn(a7, 1.0001f, 0.5f);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    float *d; hipMalloc(&d, 256 * 2048 * 8 * sizeof(float));
    const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffffff00000000ull, 0x5555555555555555ull, 0xffffull, 1ull, 0x00000000ffff0000ull, 0xffff0000ffffull};
    const char *names[] = {"all 64", "low 32", "high 32", "even lanes", "low 16", "1 lane", "lanes 16-31", "lanes 0-15 + 32-47"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int m = 0; m < 8; ++m) {
        k<<<2048 * 8, 256>>>(d, masks[m], 4000); hipDeviceSynchronize();
        hipEventRecord(e0); k<<<2048 * 8, 256>>>(d, masks[m], 4000); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-20s %8.3f ms\n", names[m], ms);
    }
    return 0;
}
