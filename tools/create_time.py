#!/usr/bin/env python3
"""ON THE GPU BOX: wall time of jtx_mi_scene_create for the 262 k-triangle atrium (JTX_TRACE_CREATE=1: the stages on stderr)
and of a device refit."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jtx_pathtracer_amd as jtx
data = jtx.scenes.atrium(262144)
t = time.time(); desc = data.to_desc(); print(f"to_desc (python): {(time.time() - t) * 1000:.1f} ms")
lib = jtx._capi.load()
for i in range(3):
    h = C.c_void_p()
    t = time.time(); jtx._capi.check(lib.jtx_mi_scene_create(C.byref(desc), C.byref(h))); dt = time.time() - t
    print(f"jtx_mi_scene_create {i}: {dt * 1000:.1f} ms", flush=True)
    lib.jtx_mi_scene_destroy(h)
