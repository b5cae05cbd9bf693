#!/usr/bin/env python3
"""ON THE GPU BOX: wall time of jtx_mi_scene_refit on the 262 k-triangle atrium, first call of the process and steady state
(JTX_TRACE_CREATE=1 prints the stages)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jtx_pathtracer_amd as jtx
data = jtx.scenes.atrium(262144)
sc = jtx.Scene(data); sc.buildBVH()
m = np.eye(4, dtype=np.float32); m[0, 3] = 3.0
for i in range(4):
    sc.setTransform(0, m)
    t = time.perf_counter(); sc.refit(); dt = time.perf_counter() - t
    print(f"edit + refit {i}: {dt * 1e3:.2f} ms", flush=True)
