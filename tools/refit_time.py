import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import jtx_pathtracer_amd as jtx
data = jtx.scenes.atrium(262144)
sc = jtx.Scene(data); sc.buildBVH()
m = np.eye(4, dtype=np.float32); m[0, 3] = 3.0
for i in range(4):
    sc.setTransform(0, m); t = time.perf_counter(); sc.refit(); print(f"refit {i}: {(time.perf_counter() - t) * 1e3:.2f} ms", flush=True)
