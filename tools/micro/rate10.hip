// micro-benchmark 10 (round 5): one leaf box of phase A exactly as the compiler emits it (order AND register numbers, from the listing of
// k_render_paths<3, 1, 256>) against the same instructions with registers renamed so that no instruction reads two operands from one
// bank (bank = register number mod 4).  8 waves per SIMD, 8 boxes per trip, no memory.
#include <hip/hip_runtime.h>
#include <cstdio>
// ray: o = v20 v21 v25, ip = v15 v17 v46, in = v47 v48 v49; pos word v8, mask v9
#define REAL \
    "v_sub_f32_e32 v24, s5, v20\nv_cmp_le_f32_e32 vcc, v11, v10\nv_sub_f32_e32 v11, s4, v20\nv_mul_f32_e32 v38, v47, v24\nv_sub_f32_e32 v34, s7, v21\nv_fmac_f32_e32 v38, v11, v15\n" \
    "v_mul_f32_e32 v11, v47, v11\nv_sub_f32_e32 v28, s6, v21\nv_fmac_f32_e32 v11, v24, v15\nv_mul_f32_e32 v24, v48, v34\nv_sub_f32_e32 v37, s3, v25\nv_fmac_f32_e32 v24, v28, v17\n" \
    "v_mul_f32_e32 v28, v48, v28\nv_sub_f32_e32 v36, s2, v25\nv_fmac_f32_e32 v28, v34, v17\nv_mul_f32_e32 v34, v49, v37\nv_fmac_f32_e32 v34, v36, v46\nv_mul_f32_e32 v36, v49, v36\n" \
    "v_fmac_f32_e32 v36, v37, v46\nv_max_f32_e32 v34, 0x3a83126f, v34\nv_max3_f32 v24, v38, v24, v34\nv_min3_f32 v11, v11, v28, v36\nv_cndmask_b32_e64 v10, 0, 1, vcc\n" \
    "v_lshl_or_b32 v9, v10, v8, v9\n"
// the same stream, renamed: o = v20 v21 v22 (banks 0 1 2), ip = v13 v14 v15 (1 2 3), in = v16 v17 v18 (0 1 2); temporaries chosen per instruction
//   a/b per axis: x: a = v27 (3) b = v30 (2); y: a = v31 (3) b = v32 (0); z: a = v33 (1) b = v35 (3)
//   near/far: x: n = v40 (0) f = v41 (1); y: n = v42 (2) f = v43 (3); z: n = v44 (0) f = v45 (1)
#define RENAMED \
    "v_sub_f32_e32 v30, s5, v20\nv_cmp_le_f32_e32 vcc, v41, v10\nv_sub_f32_e32 v27, s4, v20\nv_mul_f32_e32 v40, v17, v30\nv_sub_f32_e32 v32, s7, v21\nv_fmac_f32_e32 v40, v27, v13\n" \
    "v_mul_f32_e32 v41, v16, v27\nv_sub_f32_e32 v31, s6, v21\nv_fmac_f32_e32 v41, v30, v15\nv_mul_f32_e32 v42, v16, v32\nv_sub_f32_e32 v35, s3, v22\nv_fmac_f32_e32 v42, v31, v13\n" \
    "v_mul_f32_e32 v43, v18, v31\nv_sub_f32_e32 v33, s2, v22\nv_fmac_f32_e32 v43, v32, v14\nv_mul_f32_e32 v44, v18, v35\nv_fmac_f32_e32 v44, v33, v15\nv_mul_f32_e32 v45, v16, v33\n" \
    "v_fmac_f32_e32 v45, v35, v14\nv_max_f32_e32 v44, 0x3a83126f, v44\nv_max3_f32 v46, v40, v42, v44\nv_min3_f32 v47, v41, v43, v45\nv_cndmask_b32_e64 v10, 0, 1, vcc\n" \
    "v_lshl_or_b32 v9, v10, v8, v9\n"
#define CLOB "v9", "v10", "v11", "v24", "v27", "v28", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc"
template <int OP>
__global__ void __launch_bounds__(256, 8) k(float *out, int iters) {
    asm volatile("v_mov_b32 v20, 1.0\nv_mov_b32 v21, 2.0\nv_mov_b32 v25, 0.5\nv_mov_b32 v22, 0.5\nv_mov_b32 v15, 4.0\nv_mov_b32 v17, -2.0\nv_mov_b32 v46, 1.0\nv_mov_b32 v47, 0\nv_mov_b32 v48, 0\nv_mov_b32 v49, 0\n"
                 "v_mov_b32 v13, 4.0\nv_mov_b32 v14, -2.0\nv_mov_b32 v16, 0\nv_mov_b32 v18, 0\nv_mov_b32 v8, 3\nv_mov_b32 v9, 0\nv_mov_b32 v10, 0\nv_mov_b32 v11, 0\nv_mov_b32 v41, 0\n"
                 "s_mov_b32 s2, 1.0\ns_mov_b32 s3, 2.0\ns_mov_b32 s4, 0.5\ns_mov_b32 s5, 4.0\ns_mov_b32 s6, -2.0\ns_mov_b32 s7, 1.0"
                 ::: "v8", "v9", "v10", "v11", "v13", "v14", "v15", "v16", "v17", "v18", "v20", "v21", "v22", "v25", "v41", "v46", "v47", "v48", "v49", "s2", "s3", "s4", "s5", "s6", "s7");
    const long long c_0 = clock64(), w_0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) asm volatile(REAL REAL REAL REAL REAL REAL REAL REAL ::: CLOB);
        if (OP == 1) asm volatile(RENAMED RENAMED RENAMED RENAMED RENAMED RENAMED RENAMED RENAMED ::: CLOB);
        // 2 / 3: the loop body 4 x / 16 x as long (32 / 128 boxes = 768 / 3 072 instructions per trip: does straight-line code of phase A's length run at the same rate?)
#define R8 REAL REAL REAL REAL REAL REAL REAL REAL
        if (OP == 2) asm volatile(R8 R8 R8 R8 ::: CLOB);
        if (OP == 3) asm volatile(R8 R8 R8 R8 R8 R8 R8 R8 R8 R8 R8 R8 R8 R8 R8 R8 ::: CLOB);
    }
    const long long c_1 = clock64(), w_1 = wall_clock64();
    if (blockIdx.x == 1000 && threadIdx.x == 0) { ((long long *) out)[0] = c_1 - c_0; ((long long *) out)[1] = w_1 - w_0; }
}
static double g_ghz[4];
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
    float *d; (void) hipMalloc(&d, 4096);
    const int it = 2000;
    const char *names[] = {"the compiler's box (its registers)", "the same box, registers renamed", "the compiler's box, 32 per trip", "the compiler's box, 128 per trip"};
    const double per[4] = {8, 8, 32, 128};
    float ms[4] = {run<0>(d, it), run<1>(d, it), run<2>(d, it / 4), run<3>(d, it / 16)};
    for (int i = 0; i < 4; ++i) {
        const double cyc = ms[i] * 1e6 * g_ghz[i] / (8.0 * 8.0 * it);
        printf("%-42s 24 instructions  %6.1f cycles per box  (%.2f per instruction) at %.2f GHz\n", names[i], cyc, cyc / 24, g_ghz[i]);
    }
    return 0;
}
