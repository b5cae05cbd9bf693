import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jtx_pathtracer_amd as jtx
data = jtx.scenes.cornell(); sc = jtx.Scene(data); sc.buildBVH()
cam = jtx.StaticCamera(1920, 1080, data.camera, 8, 8, 8)
cam.samplesPerPass_ = 1
for _ in range(2):
    t = time.perf_counter(); cam.render(sc, progress=lambda c, t: None); dt = time.perf_counter() - t
print(f"{dt * 1e3:.2f} ms per frame")
