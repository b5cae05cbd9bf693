// micro-benchmark 9 (round 5): what one leaf box of phase A costs a SIMD, by variant, in the real MIX of instruction classes (the
// single-class rates of rate.hip .. rate8.hip do not add up: the replayed phase A of round 4 -- 581 "4.4-cycle" + 187 "2.4-cycle"
// instructions -- takes 2 576 cycles, the round-5 one -- 373 + 343 -- takes 2 560).  Eight boxes per trip, registers as the compiler
// assigns them in the kernel (distinct banks where it does), 8 waves per SIMD, no memory.
//   0  round 4: 6 v_sub (SGPR plane) + 6 v_mul + 3 v_min + 3 v_max + v_max literal + v_max3 + v_min3 + v_cmp + v_cndmask + v_lshl_or
//   1  round 5: 6 v_sub (SGPR) + 6 v_mul + 6 v_fmac + v_max literal + v_max3 + v_min3 + v_cmp + v_cndmask + v_lshl_or
//   2  as 1 with the planes in VGPRs (what the SGPR operand costs in the mix)
//   3  as 1 without the mask instructions (v_cmp only)
//   4  only the 6 v_sub (SGPR)            5  only the 6 v_mul + 6 v_fmac           6  only v_max literal + v_max3 + v_min3 + v_cmp
//   7  as 1 with v_fma_f32 (VOP3, separate destination) instead of v_mul + v_fmac pairs writing in place
#include <hip/hip_runtime.h>
#include <cstdio>
#define SUBS(P) "v_sub_f32 v10, " P "0, v0\nv_sub_f32 v11, " P "1, v0\nv_sub_f32 v12, " P "2, v1\nv_sub_f32 v13, " P "3, v1\nv_sub_f32 v14, " P "4, v2\nv_sub_f32 v15, " P "5, v2\n"
#define MULS "v_mul_f32 v16, v3, v10\nv_mul_f32 v17, v3, v11\nv_mul_f32 v18, v4, v12\nv_mul_f32 v19, v4, v13\nv_mul_f32 v20, v5, v14\nv_mul_f32 v21, v5, v15\n"
#define MINMAX "v_min_f32 v22, v16, v17\nv_min_f32 v23, v18, v19\nv_min_f32 v24, v20, v21\nv_max_f32 v24, 0x3a83126f, v24\nv_max_f32 v16, v16, v17\nv_max_f32 v18, v18, v19\nv_max_f32 v20, v20, v21\n"
#define SEL "v_mul_f32 v16, v6, v11\nv_fmac_f32 v16, v10, v3\nv_mul_f32 v17, v6, v10\nv_fmac_f32 v17, v11, v3\nv_mul_f32 v18, v7, v13\nv_fmac_f32 v18, v12, v4\n" \
            "v_mul_f32 v19, v7, v12\nv_fmac_f32 v19, v13, v4\nv_mul_f32 v20, v8, v15\nv_fmac_f32 v20, v14, v5\nv_mul_f32 v21, v8, v14\nv_fmac_f32 v21, v15, v5\n"
#define SELFMA "v_mul_f32 v22, v6, v11\nv_fma_f32 v16, v10, v3, v22\nv_mul_f32 v23, v6, v10\nv_fma_f32 v17, v11, v3, v23\nv_mul_f32 v24, v7, v13\nv_fma_f32 v18, v12, v4, v24\n" \
               "v_mul_f32 v25, v7, v12\nv_fma_f32 v19, v13, v4, v25\nv_mul_f32 v26, v8, v15\nv_fma_f32 v20, v14, v5, v26\nv_mul_f32 v27, v8, v14\nv_fma_f32 v21, v15, v5, v27\n"
#define RED_OLD "v_max3_f32 v22, v22, v23, v24\nv_min3_f32 v16, v16, v18, v20\nv_cmp_le_f32 vcc, v22, v16\n"
#define RED_NEW "v_max_f32 v20, 0x3a83126f, v20\nv_max3_f32 v16, v16, v18, v20\nv_min3_f32 v17, v17, v19, v21\nv_cmp_le_f32 vcc, v16, v17\n"
#define MASK "v_cndmask_b32 v28, 0, 1, vcc\nv_lshl_or_b32 v9, v28, v29, v9\n"
// 8: the round-5 box in the ORDER the compiler emits it (every v_mul two instructions behind the v_sub it reads, every v_fmac right behind its v_mul)
#define INTERLEAVED "v_sub_f32 v11, s21, v0\nv_sub_f32 v10, s20, v0\nv_mul_f32 v16, v6, v11\nv_sub_f32 v13, s23, v1\nv_fmac_f32 v16, v10, v3\nv_mul_f32 v17, v6, v10\n" \
    "v_sub_f32 v12, s22, v1\nv_fmac_f32 v17, v11, v3\nv_mul_f32 v18, v7, v13\nv_sub_f32 v15, s25, v2\nv_fmac_f32 v18, v12, v4\nv_mul_f32 v19, v7, v12\nv_sub_f32 v14, s24, v2\n" \
    "v_fmac_f32 v19, v13, v4\nv_mul_f32 v20, v8, v15\nv_fmac_f32 v20, v14, v5\nv_mul_f32 v21, v8, v14\nv_fmac_f32 v21, v15, v5\n"
// 9: the clustered order with every operand of an instruction in the same register bank (v0 v4 v8 ... : bank = number mod 4)
#define BANKED "v_sub_f32 v12, s20, v0\nv_sub_f32 v16, s21, v0\nv_sub_f32 v20, s22, v4\nv_sub_f32 v24, s23, v4\nv_sub_f32 v28, s24, v8\nv_sub_f32 v32, s25, v8\n" \
    "v_mul_f32 v36, v40, v16\nv_fmac_f32 v36, v12, v44\nv_mul_f32 v48, v40, v12\nv_fmac_f32 v48, v16, v44\nv_mul_f32 v52, v56, v24\nv_fmac_f32 v52, v20, v60\n" \
    "v_mul_f32 v64, v56, v20\nv_fmac_f32 v64, v24, v60\nv_mul_f32 v68, v40, v32\nv_fmac_f32 v68, v28, v44\nv_mul_f32 v12, v40, v28\nv_fmac_f32 v12, v32, v44\n" \
    "v_max_f32 v68, 0x3a83126f, v68\nv_max3_f32 v36, v36, v52, v68\nv_min3_f32 v48, v48, v64, v12\nv_cmp_le_f32 vcc, v36, v48\nv_cndmask_b32 v28, 0, 1, vcc\nv_lshl_or_b32 v20, v28, v24, v20\n"
#define CLOBB "v12", "v16", "v20", "v24", "v28", "v32", "v36", "v48", "v52", "v64", "v68", "vcc"
// 10: the fma select with packed fp32: [near, far] = [a, b] * [ip, ip] + [b, a] * [in, in]; (a, b) = v[10:11] .., (ip, in) = v[36:37] v[38:39] v[40:41]
#define PKSEL "v_pk_mul_f32 v[16:17], v[10:11], v[36:37] op_sel:[1,1] op_sel_hi:[0,1]\nv_pk_fma_f32 v[16:17], v[10:11], v[36:37], v[16:17] op_sel_hi:[1,0,1]\n" \
              "v_pk_mul_f32 v[18:19], v[12:13], v[38:39] op_sel:[1,1] op_sel_hi:[0,1]\nv_pk_fma_f32 v[18:19], v[12:13], v[38:39], v[18:19] op_sel_hi:[1,0,1]\n" \
              "v_pk_mul_f32 v[20:21], v[14:15], v[40:41] op_sel:[1,1] op_sel_hi:[0,1]\nv_pk_fma_f32 v[20:21], v[14:15], v[40:41], v[20:21] op_sel_hi:[1,0,1]\n"
#define CLOB "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "vcc"
#define BOX8(B) asm volatile(B B B B B B B B ::: CLOB)
template <int OP>
__global__ void __launch_bounds__(256, 8) k(float *out, int iters) {
    asm volatile("v_mov_b32 v0, 1.0\nv_mov_b32 v1, 2.0\nv_mov_b32 v2, 0.5\nv_mov_b32 v3, 4.0\nv_mov_b32 v4, -2.0\nv_mov_b32 v5, 1.0\nv_mov_b32 v6, 0\nv_mov_b32 v7, 0\nv_mov_b32 v8, 0\nv_mov_b32 v9, 0\nv_mov_b32 v29, 3\n"
                 "v_mov_b32 v36, 4.0\nv_mov_b32 v37, 0\nv_mov_b32 v38, -2.0\nv_mov_b32 v39, 0\nv_mov_b32 v40, 0\nv_mov_b32 v41, 1.0\nv_mov_b32 v30, 1.0\nv_mov_b32 v31, 2.0\nv_mov_b32 v32, 0.5\nv_mov_b32 v33, 4.0\nv_mov_b32 v34, -2.0\nv_mov_b32 v35, 1.0\n"
                 "s_mov_b32 s20, 1.0\ns_mov_b32 s21, 2.0\ns_mov_b32 s22, 0.5\ns_mov_b32 s23, 4.0\ns_mov_b32 s24, -2.0\ns_mov_b32 s25, 1.0"
                 ::: "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "s20", "s21", "s22", "s23", "s24", "s25");
    const long long c_0 = clock64(), w_0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) BOX8(SUBS("s2") MULS MINMAX RED_OLD MASK);
        if (OP == 1) BOX8(SUBS("s2") SEL RED_NEW MASK);
        if (OP == 2) BOX8(SUBS("v3") SEL RED_NEW MASK);
        if (OP == 3) BOX8(SUBS("s2") SEL RED_NEW);
        if (OP == 4) BOX8(SUBS("s2"));
        if (OP == 5) BOX8(SEL);
        if (OP == 6) BOX8(RED_NEW);
        if (OP == 7) BOX8(SUBS("s2") SELFMA RED_NEW MASK);
        if (OP == 8) BOX8(INTERLEAVED RED_NEW MASK);
        if (OP == 10) BOX8(SUBS("s2") PKSEL RED_NEW MASK);
        if (OP == 11) BOX8(PKSEL);
        if (OP == 9) asm volatile(BANKED BANKED BANKED BANKED BANKED BANKED BANKED BANKED ::: CLOBB);
    }
    const long long c_1 = clock64(), w_1 = wall_clock64();
    if (blockIdx.x == 1000 && threadIdx.x == 0) { ((long long *) out)[0] = c_1 - c_0; ((long long *) out)[1] = w_1 - w_0; }
}
static double g_ghz[16];
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
    const int it = 4000;
    const char *names[] = {"round 4 box (min / max)", "round 5 box (fma select)", "round 5 box, planes in VGPRs", "round 5 box without the mask", "6 v_sub from SGPRs", "6 v_mul + 6 v_fmac",
                           "v_max literal + v_max3 + v_min3 + v_cmp", "round 5 box with v_fma_f32 (VOP3)", "round 5 box in the compiler's order", "round 5 box, one register bank", "box with v_pk_mul_f32 + v_pk_fma_f32", "3 v_pk_mul_f32 + 3 v_pk_fma_f32"};
    const int ninstr[] = {24, 24, 24, 22, 6, 12, 4, 24, 24, 24, 18, 6};
    float ms[12] = {run<0>(d, it), run<1>(d, it), run<2>(d, it), run<3>(d, it), run<4>(d, it), run<5>(d, it), run<6>(d, it), run<7>(d, it), run<8>(d, it), run<9>(d, it), run<10>(d, it), run<11>(d, it)};
    for (int i = 0; i < 12; ++i) {
        const double cyc = ms[i] * 1e6 * g_ghz[i] / (8.0 * 8.0 * it);      // cycles of the SIMD per box of ONE wave (8 waves share it, 8 boxes per trip)
        printf("%-42s %2d instructions  %6.1f cycles per box  (%.2f per instruction) at %.2f GHz\n", names[i], ninstr[i], cyc, cyc / ninstr[i], g_ghz[i]);
    }
    return 0;
}
