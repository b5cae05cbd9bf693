#!/usr/bin/env python3
"""ON THE GPU BOX (library built with -DJTX_PROFILE_WIDE -DJTX_PROFILE_FUSE on every source): what would ONE traversal call per bounce -- the shadow
ray of the previous vertex and the extension ray walked back to back by the same lane -- save in wave iterations?  Per bounce of a wave: separate
calls cost max_lanes(steps of the extension ray) + max_lanes(steps of the shadow ray); a fused call max_lanes(shadow steps + extension steps)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jtx_pathtracer_amd as jtx
lib = jtx._capi.load()
which = sys.argv[1] if len(sys.argv) > 1 else "atrium"
data = jtx.scenes.atrium() if which == "atrium" else jtx.scenes.mixed()
sc = jtx.Scene(data); sc.buildBVH()
cam = jtx.StaticCamera(960, 540, data.camera, 4, 4, 8)
cam.render(sc, count_rays=True, integrator=1)
cam.render(sc, count_rays=False, integrator=1)
f = lib.jtx_mi_debug_fuse; f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
o = (C.c_uint64 * 3)(); assert f(sc.handle, o) == 0
sep, fused, n = [int(x) for x in o]
print(f"{which}: wave-bounces {n}; separate calls {sep / n:.2f} steps per bounce, fused {fused / n:.2f} -> {100 * (fused / sep - 1):+.1f} %")
