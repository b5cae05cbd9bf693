# diagnostic: per-rank kernel time of a 1/N tile shard of the C2 frame on ONE GPU (rehearsal of the strong-scaling run)
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jtx_pathtracer_amd as jtx
lib = jtx._capi.load()
data = jtx.scenes.cornell(); sc = jtx.Scene(data); sc.buildBVH()
W, H = 1920, 1080
cam = data.camera_desc(W, H, 8, 8, 8)
dev = torch.device("cuda", 0)
acc = torch.zeros(H * W * 3, dtype=torch.float32, device=dev); img = torch.zeros(H * W * 3, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for world in (1, 2, 4, 8):
    for rep in range(3):
        jtx.distributed.render_shard(sc, cam, 0, world, acc, img, stream=st.cuda_stream)
    torch.cuda.synchronize()
    ms = C.c_float(); n = C.c_int32(); lib.jtx_mi_kernel_time(sc.handle, C.byref(ms), C.byref(n))
    t0 = time.perf_counter()
    for rep in range(5):
        jtx.distributed.render_shard(sc, cam, 0, world, acc, img, stream=st.cuda_stream)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    lib.jtx_mi_kernel_time(sc.handle, C.byref(ms), C.byref(n))
    print(f"world {world}: rank-0 shard {ms.value / n.value:7.3f} ms GPU, {wall:7.3f} ms wall/step")
# device-side cost of the per-frame exchange (FrameGather pack on every rank, scatter on rank 0), 8-rank geometry
fg = jtx.distributed.FrameGather(W, H, 0, 8, dev)
for name, fn in (("pack", lambda: fg.pack(acc, img)), ("scatter", lambda: fg.scatter(acc, img))):
    for rep in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(20): fn()
    torch.cuda.synchronize()
    print(f"FrameGather.{name} (world 8, 1080p): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
