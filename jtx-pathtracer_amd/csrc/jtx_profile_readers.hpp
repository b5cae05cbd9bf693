// jtx_profile_readers.hpp -- host side of the diagnostic builds (jtx_profile.hpp): the jtx_mi_debug_* entry points that read the extra
// tallies back from the scene's counter block.  Included by jtx_capi.hip inside its extern "C" block, after jtx_mi_scene is defined;
// expands to NOTHING in the product build (none of these symbols is in include/jtx_mi.h or in the shipped library).
#ifdef JTX_PROFILE_UTIL
int jtx_mi_debug_util(jtx_mi_scene *s, unsigned long long *out3) {     // diagnostic builds only
    if (!s || !s->counters.p) return 1;
    (void) hipDeviceSynchronize();
    return hipMemcpy(out3, s->counters.p + 20, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
int jtx_mi_debug_util_hist(jtx_mi_scene *s, unsigned long long *out7) {
    if (!s || !s->counters.p) return 1;
    (void) hipDeviceSynchronize();
    return hipMemcpy(out7, s->counters.p + 24, 7 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif
#if defined(JTX_PROFILE_UTIL) && !defined(JTX_PROFILE_WIDE)
int jtx_mi_debug_wide_idle(jtx_mi_scene *s, unsigned long long *out4) {   // interior iterations parked / done, leaf phases walking / done
    if (!s || !s->counters.p) return 1;
    (void) hipDeviceSynchronize();
    return hipMemcpy(out4, s->counters.p + 48, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif
#ifdef JTX_PROFILE_WIDE
int jtx_mi_debug_wide(jtx_mi_scene *s, unsigned long long *out8) {     // diagnostic builds only
    if (!s || !s->counters.p) return 1;
    (void) hipDeviceSynchronize();
    return hipMemcpy(out8, s->counters.p + 24, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
int jtx_mi_debug_wide_idle(jtx_mi_scene *s, unsigned long long *out4) {   // node iterations parked / done, leaf iterations walking / done
    if (!s || !s->counters.p) return 1;
    (void) hipDeviceSynchronize();
    return hipMemcpy(out4, s->counters.p + 48, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
int jtx_mi_debug_wide_hist(jtx_mi_scene *s, unsigned long long *out7) {
    if (!s || !s->counters.p) return 1;
    (void) hipDeviceSynchronize();
    return hipMemcpy(out7, s->counters.p + 9, 7 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif
#ifdef JTX_PROFILE_TIMELINE
int jtx_mi_debug_timeline(jtx_mi_scene *s, unsigned long long *out, int n) {   // diagnostic builds only: (start, end) wall clocks per wave
    if (!s || !s->counters.p) return 1;
    (void) hipDeviceSynchronize();
    return hipMemcpy(out, s->counters.p + 64, (size_t) 2 * n * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif
#ifdef JTX_PROFILE_PHASES
int jtx_mi_debug_phases_reset(jtx_mi_scene *s) {       // diagnostic builds only
    if (!s || !s->counters.p) return 1;
    (void) hipDeviceSynchronize();
    return hipMemset(s->counters.p + 16, 0, 7 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
int jtx_mi_debug_phases(jtx_mi_scene *s, unsigned long long *out6) {   // diagnostic builds only
    if (!s || !s->counters.p) return 1;
    (void) hipDeviceSynchronize();
    return hipMemcpy(out6, s->counters.p + 16, 7 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;   // [6]: hand-out (timed kernel only)
}
#endif
