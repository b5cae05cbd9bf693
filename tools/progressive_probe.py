# ON THE GPU BOX: one C2 frame through jtx_mi_render with a callback per pass (progressive launch): wall time, path-kernel time (HIP events)
# by samplesPerPass; env JTX_RESOLVER_WGS / JTX_TRACE_RENDER / JTX_PROGRESSIVE_LAUNCH=0 (round 5's pass-by-pass loop) apply.
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jtx_pathtracer_amd as jtx
lib = jtx._capi.load()
wl = sys.argv[1] if len(sys.argv) > 1 else "cornell"
data = getattr(jtx.scenes, wl)() if wl != "atrium" else jtx.scenes.atrium(262144)
sc = jtx.Scene(data); sc.buildBVH()
xs, ys = (8, 8) if wl != "mixed" else (16, 8)
import numpy as np
check = os.environ.get("JTX_PROBE_CHECK", "1") != "0"      # (0: no batch launch in the process -- the rocprofv3 runs of tools/r06_collect.sh)
if check:
    batch = jtx.StaticCamera(1920, 1080, data.camera, xs, ys, 8); batch.render(sc)   # the frame in one batch launch: what every progressive film must equal
for spp_pass in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["8", "1"])]:
    cam = jtx.StaticCamera(1920, 1080, data.camera, xs, ys, 8)
    cam.samplesPerPass_ = spp_pass
    n = [0]
    cam.render(sc, progress=lambda c, t: n.__setitem__(0, n[0] + 1))      # warm
    ms = C.c_float(); nl = C.c_int32(); lib.jtx_mi_kernel_time(sc.handle, C.byref(ms), C.byref(nl))
    best, kbest = 1e9, 0
    for _ in range(4):
        n[0] = 0; t = time.perf_counter(); cam.render(sc, progress=lambda c, t: n.__setitem__(0, n[0] + 1)); dt = time.perf_counter() - t
        lib.jtx_mi_kernel_time(sc.handle, C.byref(ms), C.byref(nl))
        if dt < best: best, kbest = dt, ms.value
    same = (np.array_equal(cam.acc_.view(np.uint32), batch.acc_.view(np.uint32)) and np.array_equal(cam.img_, batch.img_)) if check else "not checked"
    print(f"{wl} samplesPerPass {spp_pass:3d}: {best * 1e3:8.2f} ms per frame, path kernel(s) {kbest:7.2f} ms, {n[0]} callbacks; film and image == the batch frame's, every pixel: {same}", flush=True)
    if same is False: sys.exit(2)
