# diagnostic: SIMD lane utilisation of the traversal loops of integrator 1 (build with -DJTX_PROFILE_UTIL)
import ctypes as C, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jtx_pathtracer_amd as jtx
lib = jtx._capi.load()
data = jtx.scenes.cornell(); sc = jtx.Scene(data); sc.buildBVH()
cam = jtx.StaticCamera(1920, 1080, data.camera, 8, 8, 8)
cam.render(sc, count_rays=True, integrator=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
f = lib.jtx_mi_debug_util; f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
out = (C.c_uint64 * 3)(); assert f(sc.handle, out) == 0
c = cam.counters
it_int, it_leaf, calls = out[0], out[1], out[2]
nodes = c["n_nodes_closest"] + c["n_nodes_any"]; tris = c["n_tri_closest"] + c["n_tri_any"]; rays = c["n_closest"] + c["n_any"]
print("wave interior iterations", it_int, "lane node visits", nodes, "utilisation %.3f" % (nodes / (64.0 * it_int)))
print("wave leaf phases", it_leaf, "lane tri tests", tris, "utilisation(approx, 1-2 tris/leaf) %.3f" % (tris / (64.0 * it_leaf)))
print("wave traverse calls", calls, "lane rays", rays, "utilisation %.3f" % (rays / (64.0 * calls)))
print("interior iterations per call %.1f, leaf phases per call %.2f" % (it_int / calls, it_leaf / calls))
print("mean node visits/ray %.1f" % (nodes / rays))
h = lib.jtx_mi_debug_util_hist; h.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
hv = (C.c_uint64 * 7)(); assert h(sc.handle, hv) == 0
tot = sum(int(x) for x in hv)
print("interior iterations by walking lanes  1-2 3-4 5-8 9-16 17-32 33-48 49-64:", " ".join(f"{int(x) / tot:.3f}" for x in hv))
g = lib.jtx_mi_debug_wide_idle; g.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
iv = (C.c_uint64 * 4)(); assert g(sc.handle, iv) == 0
npk, nd, lw, ld = [int(x) for x in iv]
print("interior iterations, lane share: walking %.3f parked %.3f done %.3f" % (nodes / (64.0 * it_int), npk / (64.0 * it_int), nd / (64.0 * it_int)))
print("leaf phases, lane share: walking %.3f done %.3f" % (lw / (64.0 * it_leaf), ld / (64.0 * it_leaf)))
