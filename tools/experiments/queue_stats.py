# diagnostic: where k_render_queue spends its time (build every source with -DJTX_PROFILE_QUEUE)
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jtx_pathtracer_amd as jtx
lib = jtx._capi.load()
which = sys.argv[1] if len(sys.argv) > 1 else "atrium"
data = jtx.scenes.atrium() if which == "atrium" else jtx.scenes.mixed()
sc = jtx.Scene(data); sc.buildBVH()
W, H, xs, ys = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1920, 1080, 4, 4)
cam = jtx.StaticCamera(W, H, data.camera, xs, ys, 8)
cam.render(sc, count_rays=True, integrator=1)        # allocates + zeroes the counter block
c = cam.counters
cam.render(sc, count_rays=False, integrator=1)
f = lib.jtx_mi_debug_queue; f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
o = (C.c_uint64 * 14)(); assert f(sc.handle, o) == 0
rounds, tsh, ttr, nit, nst, lit, lst, refills, rays, fin, wait, parked, sslots, sshaded = [int(x) for x in o]
print(f"rays {rays} (counted {c['n_closest'] + c['n_any']}), wave rounds {rounds}, rays per wave round {rays / rounds:.1f}")
print(f"clocks: shade {tsh / (tsh + ttr):.3f} trace {ttr / (tsh + ttr):.3f}; per round shade {tsh / rounds:.0f} trace {ttr / rounds:.0f}")
print(f"node loop: {nit / rounds:.1f} iterations per round, {nst / rays:.2f} steps per ray; lane share walking {nst / (64.0 * nit):.3f} "
      f"parked {parked / (64.0 * nit):.3f} waiting for a refill {wait / (64.0 * nit):.3f} out of rays {fin / (64.0 * nit):.3f}")
print(f"leaf steps: {lit / rounds:.1f} iterations per round, {lst / rays:.2f} per ray, lane share {lst / (64.0 * lit):.3f}; refill blocks per round {refills / rounds:.1f}")
print(f"shade phase: {sslots / rounds:.2f} slot steps per round, lane share shading {sshaded / (64.0 * max(1, sslots)):.3f}")
print(f"clocks per node iteration (trace phase / node iterations): {ttr / nit:.0f}")
