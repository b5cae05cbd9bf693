// micro-benchmark 11 (round 5): WHICH pipe does an instruction run on?  Four of it alone, four v_max3_f32 (half-rate pipe) alone, four
// v_fma_f32 (full-rate pipe) alone, and the mixes: on different pipes a mix takes the larger of the parts, on the same pipe their sum.
// 8 waves per SIMD, independent instructions, no memory.
#include <hip/hip_runtime.h>
#include <cstdio>
#define MAX3 "v_max3_f32 v20, v0, v1, v2\nv_max3_f32 v21, v1, v2, v3\nv_max3_f32 v22, v2, v3, v4\nv_max3_f32 v23, v3, v4, v5\n"
#define FMA4 "v_fma_f32 v24, v0, v1, v2\nv_fma_f32 v25, v1, v2, v3\nv_fma_f32 v26, v2, v3, v4\nv_fma_f32 v27, v3, v4, v5\n"
#define X2(S) S " v28, v6, v7\n" S " v29, v7, v8\n" S " v30, v8, v9\n" S " v31, v9, v6\n"
#define X3(S) S " v28, v6, v7, v8\n" S " v29, v7, v8, v9\n" S " v30, v8, v9, v6\n" S " v31, v9, v6, v7\n"
#define X1(S) S " v28, v6\n" S " v29, v7\n" S " v30, v8\n" S " v31, v9\n"
#define CNDM "v_cndmask_b32 v28, v6, v7, vcc\nv_cndmask_b32 v29, v7, v8, vcc\nv_cndmask_b32 v30, v8, v9, vcc\nv_cndmask_b32 v31, v9, v6, vcc\n"
#define CLOB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31"
#define RUN(B) asm volatile(B B B B ::: CLOB)
template <int OP, int MIX>
__global__ void __launch_bounds__(256, 8) k(float *out, int iters) {
    asm volatile("v_mov_b32 v0, 1.0\nv_mov_b32 v1, 2.0\nv_mov_b32 v2, 0.5\nv_mov_b32 v3, 4.0\nv_mov_b32 v4, -2.0\nv_mov_b32 v5, 1.0\nv_mov_b32 v6, 3\nv_mov_b32 v7, 5\nv_mov_b32 v8, 7\nv_mov_b32 v9, 9\ns_mov_b64 vcc, 0x5555\ns_mov_b64 s[20:21], 0x3333"
                 ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "vcc", "s20", "s21", "s22", "s23");
    const long long c_0 = clock64(), w_0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#define CASE(N, X) if (OP == N) { if (MIX == 0) RUN(X); if (MIX == 1) RUN(MAX3 X); if (MIX == 2) RUN(FMA4 X); }
        CASE(0, X2("v_and_b32"))  CASE(1, X2("v_xor_b32"))  CASE(2, X2("v_or_b32"))  CASE(3, X2("v_add_u32"))  CASE(4, X3("v_lshl_add_u32"))
        CASE(5, X3("v_mad_u32_u24"))  CASE(6, X1("v_mov_b32"))  CASE(7, CNDM)  CASE(8, X3("v_bfi_b32"))  CASE(9, X2("v_lshlrev_b32"))
        CASE(10, X1("v_cvt_f32_ubyte1"))  CASE(11, X3("v_perm_b32"))  CASE(12, X2("v_sub_f32"))  CASE(13, X3("v_min3_f32"))
        CASE(14, X2("v_mul_u32_u24"))  CASE(15, X3("v_add3_u32"))  CASE(16, X3("v_and_or_b32"))  CASE(17, X2("v_ashrrev_i32"))
        CASE(18, X1("v_rcp_f32"))  CASE(19, X2("v_mul_f32"))
        CASE(20, "v_swap_b32 v28, v29\nv_swap_b32 v30, v31\nv_swap_b32 v28, v30\nv_swap_b32 v29, v31\n")
        CASE(21, "v_cndmask_b32_e64 v28, v6, v7, s[20:21]\nv_cndmask_b32_e64 v29, v7, v8, s[20:21]\nv_cndmask_b32_e64 v30, v8, v9, s[20:21]\nv_cndmask_b32_e64 v31, v9, v6, s[20:21]\n")
        CASE(22, "s_and_saveexec_b64 s[22:23], s[20:21]\nv_swap_b32 v28, v29\nv_swap_b32 v30, v31\ns_mov_b64 exec, s[22:23]\ns_and_saveexec_b64 s[22:23], s[20:21]\nv_swap_b32 v28, v30\nv_swap_b32 v29, v31\ns_mov_b64 exec, s[22:23]\n")
    }
    const long long c_1 = clock64(), w_1 = wall_clock64();
    if (blockIdx.x == 1000 && threadIdx.x == 0) { ((long long *) out)[0] = c_1 - c_0; ((long long *) out)[1] = w_1 - w_0; }
}
template <int OP, int MIX> double run(float *d, int iters) {
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    k<OP, MIX><<<256 * 8, 256>>>(d, iters); (void) hipDeviceSynchronize();
    (void) hipEventRecord(e0); k<OP, MIX><<<256 * 8, 256>>>(d, iters); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; (void) hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const double ghz = (double) h[0] / ((double) h[1] * 10.0);
    return ms * 1e6 * ghz / (8.0 * 4.0 * iters);          // SIMD cycles per group (4 of X [+ 4 of the partner]) of one wave
}
template <int OP> void row(float *d, const char *name) {
    const int it = 4000;
    const double a = run<OP, 0>(d, it), b = run<OP, 1>(d, it), c = run<OP, 2>(d, it);
    printf("%-18s 4 alone %5.1f   + 4 v_max3_f32 (17.7 alone) %5.1f -> %s   + 4 v_fma_f32 (8.6 alone) %5.1f -> %s\n", name, a, b,
           b < 0.8 * (a + 17.7) ? "side by side" : "same pipe", c, c < 0.8 * (a + 8.6) ? "side by side" : "adds up");
}
int main() {
    float *d; (void) hipMalloc(&d, 4096);
    row<0>(d, "v_and_b32"); row<1>(d, "v_xor_b32"); row<2>(d, "v_or_b32"); row<3>(d, "v_add_u32"); row<4>(d, "v_lshl_add_u32"); row<5>(d, "v_mad_u32_u24");
    row<6>(d, "v_mov_b32"); row<7>(d, "v_cndmask_b32"); row<8>(d, "v_bfi_b32"); row<9>(d, "v_lshlrev_b32"); row<10>(d, "v_cvt_f32_ubyte1"); row<11>(d, "v_perm_b32");
    row<12>(d, "v_sub_f32"); row<13>(d, "v_min3_f32"); row<14>(d, "v_mul_u32_u24"); row<15>(d, "v_add3_u32"); row<16>(d, "v_and_or_b32"); row<17>(d, "v_ashrrev_i32");
    row<18>(d, "v_rcp_f32"); row<19>(d, "v_mul_f32"); row<20>(d, "v_swap_b32"); row<21>(d, "v_cndmask_e64"); row<22>(d, "v_swap under exec");
    return 0;
}
