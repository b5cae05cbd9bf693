#!/usr/bin/env python3
"""ON THE GPU BOX: render N uncounted frames of a bench workload (the exact launch bench.py times) and print the
HIP-event kernel time -- the lean program rocprofv3 passes wrap (tools/pmc_collect.sh)."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (WORKLOADS)

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cornell_1920x1080_64spp_d8")
ap.add_argument("--frames", type=int, default=1)
ap.add_argument("--warmup", type=int, default=0)
ap.add_argument("--atrium-tris", type=int, default=262144)
a = ap.parse_args()
import torch
import jtx_pathtracer_amd as jtx
lib = jtx._capi.load()
jtx._capi.check(lib.jtx_mi_set_device(0))
dev = torch.device("cuda", 0)
factory, W, H, xs, ys, depth = bench.WORKLOADS[a.workload]
data = getattr(jtx.scenes, factory)(a.atrium_tris) if factory == "atrium" else getattr(jtx.scenes, factory)()
scene = jtx.Scene(data); scene.buildBVH()
cam = data.camera_desc(W, H, xs, ys, depth)
acc = torch.zeros(H * W * 3, dtype=torch.float32, device=dev); img = torch.zeros(H * W * 3, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
ms = C.c_float(); nl = C.c_int32()
for i in range(a.warmup + a.frames):
    if i == a.warmup:
        torch.cuda.synchronize(); lib.jtx_mi_kernel_time(scene.handle, C.byref(ms), C.byref(nl))
    jtx.distributed.render_shard(scene, cam, 0, 1, acc, img, stream=st.cuda_stream)
torch.cuda.synchronize()
jtx._capi.check(lib.jtx_mi_kernel_time(scene.handle, C.byref(ms), C.byref(nl)))
print(f"{a.workload}: {ms.value / max(1, nl.value):.3f} ms/frame kernel ({nl.value} frames)", flush=True)
