# ON THE GPU BOX: wall time of one C2 frame through jtx_mi_render (host buffers, callback per pass) by samplesPerPass
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jtx_pathtracer_amd as jtx
data = jtx.scenes.cornell(); sc = jtx.Scene(data); sc.buildBVH()
for spp_pass in (64, 8, 1):
    cam = jtx.StaticCamera(1920, 1080, data.camera, 8, 8, 8)
    cam.samplesPerPass_ = spp_pass
    n = [0]
    cam.render(sc, progress=lambda c, t: n.__setitem__(0, n[0] + 1))      # warm
    best = 1e9
    for _ in range(3):
        n[0] = 0; t = time.perf_counter(); cam.render(sc, progress=lambda c, t: n.__setitem__(0, n[0] + 1)); best = min(best, time.perf_counter() - t)
    print(f"samplesPerPass {spp_pass:3d}: {best * 1e3:8.2f} ms per 64-spp frame, {n[0]} passes", flush=True)
cam = jtx.StaticCamera(1920, 1080, data.camera, 8, 8, 8)
cam.render(sc); t = time.perf_counter(); cam.render(sc); print(f"no callback      : {(time.perf_counter() - t) * 1e3:8.2f} ms")
