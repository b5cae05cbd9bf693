# diagnostic: lane utilisation of the wide (8-ary) traversal of integrator 1 (build with -DJTX_PROFILE_WIDE)
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jtx_pathtracer_amd as jtx
lib = jtx._capi.load()
which = sys.argv[1] if len(sys.argv) > 1 else "atrium"
data = jtx.scenes.atrium() if which == "atrium" else jtx.scenes.mixed()
sc = jtx.Scene(data); sc.buildBVH()
W, H, xs, ys = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (960, 540, 4, 4)
cam = jtx.StaticCamera(W, H, data.camera, xs, ys, 8)
cam.render(sc, count_rays=True, integrator=1)        # allocates + zeroes the counter block; binary records
c = cam.counters
cam.render(sc, count_rays=False, integrator=1)       # wide nodes; the diagnostic counters accumulate
f = lib.jtx_mi_debug_wide; f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
o = (C.c_uint64 * 8)(); assert f(sc.handle, o) == 0
calls, nit, nst, lit, lst, tris, pops, fetch = [int(x) for x in o]
rays = c["n_closest"] + c["n_any"]
print(f"rays {rays}, wave traverse calls {calls} -> ray slots used {rays / (64.0 * calls):.3f}")
print(f"node phase: {nit / calls:.1f} wave iterations/call, lane steps/ray {nst / rays:.2f} (fetches {fetch / rays:.2f}), utilisation {nst / (64.0 * nit):.3f}")
print(f"leaf phase: {lit / calls:.1f} wave iterations/call, lane steps/ray {lst / rays:.2f}, utilisation {lst / (64.0 * lit):.3f}")
print(f"binary node visits/ray {(c['n_nodes_closest'] + c['n_nodes_any']) / rays:.1f}, tri tests/ray {(c['n_tri_closest'] + c['n_tri_any']) / rays:.2f}")
h = lib.jtx_mi_debug_wide_hist; h.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
hv = (C.c_uint64 * 7)(); assert h(sc.handle, hv) == 0
tot = sum(int(x) for x in hv)
print("node iterations by walking lanes  1-2 3-4 5-8 9-16 17-32 33-48 49-64:", " ".join(f"{int(x) / tot:.3f}" for x in hv))
g = lib.jtx_mi_debug_wide_idle; g.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
iv = (C.c_uint64 * 4)(); assert g(sc.handle, iv) == 0
npk, nd, lw, ld = [int(x) for x in iv]
print(f"node iterations, lane share: walking {nst / (64.0 * nit):.3f} parked {npk / (64.0 * nit):.3f} done {nd / (64.0 * nit):.3f} (rest: lanes without a path)")
print(f"leaf iterations, lane share: on a leaf {lst / (64.0 * lit):.3f} walking {lw / (64.0 * lit):.3f} done {ld / (64.0 * lit):.3f}")
import json
json.dump({"node_steps_per_ray": nst / rays, "leaf_steps_per_ray": lst / rays, "tri_tests_per_ray": (c['n_tri_closest'] + c['n_tri_any']) / rays,
           "node_iterations_per_call": nit / calls, "node_lane_share": {"walking": nst / (64.0 * nit), "parked": npk / (64.0 * nit), "done": nd / (64.0 * nit)},
           "frame": f"{W}x{H}x{xs * ys}spp", "rays": rays}, open(f"gpurun_out/wide_stats_{which}.json", "w"))
