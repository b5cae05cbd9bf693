#!/usr/bin/env python3
"""ON THE GPU BOX: wall time of jtx_mi_scene_rebuild (Scene::rebuildBVH on the device) for the 262 k-triangle atrium, the tree
against the host build's, and the C3 frame time on the device-built structures (their 8-ary nodes are laid out level by level)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jtx_pathtracer_amd as jtx
import bench
lib = jtx._capi.load()
data = jtx.scenes.atrium(262144)
sc = jtx.Scene(data); sc.buildBVH()
n0, r0 = sc.bvh()
W, H, xs, ys, depth = 1920, 1080, 8, 8, 8
cam = data.camera_desc(W, H, xs, ys, depth)
dev = torch.device("cuda", 0)
acc = torch.zeros(H * W * 3, dtype=torch.float32, device=dev); img = torch.zeros(H * W * 3, dtype=torch.uint8, device=dev)
ms = C.c_float(); nl = C.c_int32()
def frames(n, film=True):
    st = torch.cuda.current_stream().cuda_stream
    jtx.distributed.render_shard(sc, cam, 0, 1, acc, img, stream=st); torch.cuda.synchronize()
    lib.jtx_mi_kernel_time(sc.handle, C.byref(ms), C.byref(nl))
    for _ in range(n):
        jtx.distributed.render_shard(sc, cam, 0, 1, acc, img, stream=st)
    torch.cuda.synchronize(); lib.jtx_mi_kernel_time(sc.handle, C.byref(ms), C.byref(nl))
    return ms.value / max(1, nl.value), (acc.cpu().numpy().copy() if film else None)
if "--reserve" in sys.argv:
    t = time.perf_counter(); sc.reserveRebuild(); print(f"jtx_mi_scene_reserve_rebuild: {(time.perf_counter() - t) * 1e3:.2f} ms", flush=True)
t_host, film_host = frames(3)
print(f"C3 frame on the host-built scene: {t_host:.2f} ms", flush=True)
# as in the edit loop (display.cpp:893-905: render, edit, rebuild, render ...): every rebuild follows a frame.  (No film copy in between: torch's
# PAGEABLE 25 MB device-to-host copy makes the next command on another stream wait ~20 ms now and then -- tools/rebuild_probe.py -- which is the
# harness, not the builder.)
for i in range(4):
    frames(0, film=False)
    t = time.perf_counter(); sc.rebuildBVHOnDevice(); dt = time.perf_counter() - t
    print(f"jtx_mi_scene_rebuild {i}: {dt * 1e3:.2f} ms", flush=True)
n1, r1 = sc.bvh()
same = len(n0) == len(n1) and (n0["pmin"] == n1["pmin"]).all() and (n0["pmax"] == n1["pmax"]).all() and (n0["offset"] == n1["offset"]).all() \
    and (n0["num_prims"] == n1["num_prims"]).all()
print("tree equals the host build's:", bool(same), len(n1), "nodes; info", sc.info())
t_dev, film_dev = frames(3)
print(f"C3 frame on the device-built scene: {t_dev:.2f} ms; film bit-equal: {bool((film_host.view(np.uint32) == film_dev.view(np.uint32)).all())}")
# an edit + rebuild, against refit
m = np.eye(4, dtype=np.float32); m[0, 3] = 3.0
for what in ("refit (first call of the process)", "refit", "refit", "rebuild", "rebuild"):
    sc.setTransform(0, m)
    t = time.perf_counter(); (sc.refit() if what.startswith("refit") else sc.rebuildBVHOnDevice()); dt = time.perf_counter() - t
    print(f"edit + {what}: {dt * 1e3:.2f} ms")
# where do the two films differ?  (ties inside multi-primitive leaves are the only licensed difference)
d = (film_host.view(np.uint32) != film_dev.view(np.uint32)).reshape(H, W, 3).any(axis=2)
print("pixels that differ:", int(d.sum()), "of", H * W, "; leaves with more than one primitive:", int((n1["num_prims"] > 1).sum()), "of", int((n1["num_prims"] > 0).sum()))
if d.sum():
    print("max abs difference of the 64-sample sums:", float(np.abs(film_host - film_dev).max()))
fresh = jtx.Scene(jtx.scenes.atrium(262144)); fresh.buildBVH()
def counted(scene):
    st = torch.cuda.current_stream().cuda_stream
    c = data.camera_desc(480, 270, 4, 4, 8)
    a = torch.zeros(270 * 480 * 3, dtype=torch.float32, device=dev)
    jtx.distributed.render_shard(scene, c, 0, 1, a, None, stream=st, count_rays=True); torch.cuda.synchronize()
    a2 = torch.zeros(270 * 480 * 3, dtype=torch.float32, device=dev)
    jtx.distributed.render_shard(scene, c, 0, 1, a2, None, stream=st, count_rays=False); torch.cuda.synchronize()
    return a.cpu().numpy(), a2.cpu().numpy()
sc.rebuildBVHOnDevice()          # (the edit above moved mesh 0: put the transforms back first)
sc.setTransform(0, np.eye(4, dtype=np.float32)); sc.rebuildBVHOnDevice()
ch, uh = counted(fresh); cd, ud = counted(sc)
print("480x270x16: counted host vs device-built differ in", int((ch.view(np.uint32) != cd.view(np.uint32)).sum()), "words; uncounted:",
      int((uh.view(np.uint32) != ud.view(np.uint32)).sum()), "; counted vs uncounted on the device-built scene:", int((cd.view(np.uint32) != ud.view(np.uint32)).sum()))
