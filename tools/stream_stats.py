# diagnostic: block trips and lane use of k_render_stream (build with -DJTX_PROFILE_STREAM, JTX_MI_LIB=...)
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jtx_pathtracer_amd as jtx
lib = jtx._capi.load()
which = sys.argv[1] if len(sys.argv) > 1 else "cornell"
data = getattr(jtx.scenes, which)()
sc = jtx.Scene(data); sc.buildBVH()
W, H, xs, ys = (1920, 1080, 4, 4) if which == "cornell" else (960, 540, 4, 4)
cam = jtx.StaticCamera(W, H, data.camera, xs, ys, 8)
cam.render(sc, count_rays=True, integrator=1)        # allocates + zeroes the counter block
c = cam.counters
cam.render(sc, count_rays=False, integrator=1)
f = lib.jtx_mi_debug_stream; f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
o = (C.c_uint64 * 8)(); assert f(sc.handle, o) == 0
t = [int(x) for x in o[:4]]; l = [int(x) for x in o[4:]]
rays = c["n_closest"] + c["n_any"]; nodes = c["n_nodes_closest"] + c["n_nodes_any"]
print(f"{which}: rays {rays}, shading events {c['n_shade']}, binary node visits/ray {nodes / rays:.1f}")
for i, name in enumerate(("node", "leaf", "shade", "begin")):
    print(f"{name:6s} trips {t[i]:12d}  lanes/trip {l[i] / max(1, t[i]):6.2f}  trips per 64 rays {t[i] * 64.0 / rays:7.2f}")
