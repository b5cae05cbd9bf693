# diagnostic: per-phase wave-cycle shares (needs a build with -DJTX_PROFILE_PHASES)
import ctypes as C, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jtx_pathtracer_amd as jtx
lib = jtx._capi.load()
which = sys.argv[1] if len(sys.argv) > 1 else "cornell"
timed = "--timed" in sys.argv                           # the timed kernel (k_render_paths) instead of the counting one
data = getattr(jtx.scenes, which)(); sc = jtx.Scene(data); sc.buildBVH()
cam = jtx.StaticCamera(1920, 1080, data.camera, 4, 4, 8)
cam.render(sc, count_rays=True)
if timed:
    lib.jtx_mi_debug_phases_reset.argtypes = [C.c_void_p]; lib.jtx_mi_debug_phases_reset(sc.handle)
    cam.render(sc, count_rays=False, integrator=1)
f = lib.jtx_mi_debug_phases; f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
out = (C.c_uint64 * 7)()
assert f(sc.handle, out) == 0
v = list(out); tot = v[5]
names = ["closest traversal", "surface+light sample", "shadow traversal", "eval/pdf + accumulate", "bsdf sample", "TOTAL(kernel loop)"]
for n, x in zip(names, v): print(f"{n:28s} {x:16d} {100.0*x/tot:6.2f}%")
print("unaccounted (regen, clamp, loop)", 100.0*(tot-sum(v[:5]))/tot)
if timed: print(f"  of which the hand-out loop (chunk fetches, camera rays) {100.0 * v[6] / tot:6.2f}%")
