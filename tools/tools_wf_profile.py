# diagnostic: per-kernel-kind GPU time of one wavefront frame (C2 by default)
import ctypes as C, sys, os
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jtx_pathtracer_amd as jtx
lib = jtx._capi.load()
W, H, xs, ys, D = [int(x) for x in (sys.argv[2:7] if len(sys.argv) > 6 else (1920, 1080, 8, 8, 8))]
data = getattr(jtx.scenes, sys.argv[1] if len(sys.argv) > 1 else "cornell")(); sc = jtx.Scene(data); sc.buildBVH()
cam = data.camera_desc(W, H, xs, ys, D)
dev = torch.device("cuda", 0)
acc = torch.zeros(H * W * 3, dtype=torch.float32, device=dev); img = torch.zeros(H * W * 3, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for prof in (False, True, True):
    jtx.distributed.render_shard(sc, cam, 0, 1, acc, img, stream=st.cuda_stream, integrator=2, profile_kernels=prof)
    torch.cuda.synchronize()
ms = (C.c_float * 5)(); n = (C.c_int32 * 5)()
lib.jtx_mi_kernel_time_by_kind(sc.handle, ms, n)
names = ["generate", "trace closest", "shade", "trace any", "resolve"]
tot = sum(ms)
for i in range(5): print(f"{names[i]:14s} {ms[i]/2:9.3f} ms/frame  {n[i]//2:5d} launches  {100*ms[i]/tot:5.1f}%")
print("sum", tot / 2)
t = C.c_float(); l = C.c_int32(); lib.jtx_mi_kernel_time(sc.handle, C.byref(t), C.byref(l)); print("frame events", t.value / l.value)
