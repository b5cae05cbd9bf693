// jtx_progressive.hpp -- internal: ONE progressive launch of a scene (all passes of a strata range in one k_render_paths<.., PROG> launch, k_resolve_progressive
// beside it), as jtx_mi_render and jtx_mi_multi_render drive it.  Defined in jtx_capi.hip; not part of the C-ABI.
#pragma once
#include "../../include/jtx_mi.h"
#include <hip/hip_runtime.h>

struct JtxProgRun {
    int begin = 0, end = 0;          // strata of this launch
    int tick = 1;                    // strata per pass of the caller
    int spg = 1, groups = 0;         // strata per group (chunk), groups of the launch
    int resolver_wgs = 0;
    unsigned epoch = 0;              // tags the resolver's words of this launch in host memory
    bool nothing = false;            // the shard owns no tiles: nothing was launched
};
// the persistent path kernel takes this render (integrateMIS, uncounted, integrator 1) and progressive launches are not switched off
bool jtx_prog_usable(jtx_mi_scene *s, const jtx_mi_render_opts &o);
// strata ONE launch may cover under the radiance-record cap, in whole passes (throws when one pass does not fit)
long jtx_prog_span(jtx_mi_scene *s, const jtx_mi_camera_desc &cam, const jtx_mi_render_opts &o, int tick);
// enqueue the launch of strata [b0, e0) into device film d_acc / d_img on `stream` (the resolver goes on the scene's own resolver stream; `stream` then waits
// for it).  extra_leave_waves: wave slots the path grid leaves free beside the resolver's (for kernels the caller runs meanwhile: preview packs).
// locked: take the scene's mutex (callers that do not hold it).  Throws std::runtime_error.
void jtx_prog_begin(jtx_mi_scene *s, const jtx_mi_camera_desc &cam, const jtx_mi_render_opts &o, int b0, int e0, int tick, float *d_acc, unsigned char *d_img,
                    hipStream_t stream, int extra_leave_waves, bool locked, JtxProgRun &run);
// one past the last stratum of the launch that is in the film of every owned pixel (whole passes); *gave_up: the resolver ended on its bounded wait
int jtx_prog_completed(jtx_mi_scene *s, const JtxProgRun &run, bool *gave_up);
// both kernels of the scene's last progressive launch have ended
bool jtx_prog_finished(jtx_mi_scene *s);
