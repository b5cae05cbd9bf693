#!/usr/bin/env python3
"""ON THE GPU BOX: what does the first command after a rendered frame cost?  (round 4: a rebuild that follows a frame took 30 ms against 6.7 ms for one
that follows another rebuild; the trace put the difference into the first 4 KB host-to-device copy)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jtx_pathtracer_amd as jtx
lib = jtx._capi.load()
data = jtx.scenes.atrium(262144)
sc = jtx.Scene(data); sc.buildBVH(); sc.reserveRebuild()
W, H = 1920, 1080
cam = data.camera_desc(W, H, 8, 8, 8)
dev = torch.device("cuda", 0)
acc = torch.zeros(H * W * 3, dtype=torch.float32, device=dev); img = torch.zeros(H * W * 3, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def frame(): jtx.distributed.render_shard(sc, cam, 0, 1, acc, img, stream=st); torch.cuda.synchronize()
def t(f):
    a = time.perf_counter(); f(); return (time.perf_counter() - a) * 1e3
frame(); sc.rebuildBVHOnDevice()
small = torch.zeros(1024, dtype=torch.float32)
for what in ("rebuild after rebuild", "frame, rebuild", "frame, film to host, rebuild", "frame, sleep 0.2 s, rebuild", "frame, tiny torch H2D, rebuild", "frame, refit"):
    r = []
    for i in range(3):
        if what != "rebuild after rebuild": frame()
        if "film" in what: _ = acc.cpu()
        if "sleep" in what: time.sleep(0.2)
        if "tiny" in what: r0 = t(lambda: (small.to(dev), torch.cuda.synchronize()))
        if "refit" in what:
            m = np.eye(4, dtype=np.float32); m[0, 3] = 0.5 * i; sc.setTransform(0, m); r.append(t(sc.refit))
        else: r.append(t(sc.rebuildBVHOnDevice))
    print(f"{what:34s}: " + " ".join(f"{x:7.2f}" for x in r) + (" ms (tiny copy itself %.2f ms)" % r0 if "tiny" in what else " ms"), flush=True)
