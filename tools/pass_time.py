# ON THE GPU BOX: kernel time of partial-strata passes of C2 through jtx_mi_render_device (device film)
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import jtx_pathtracer_amd as jtx
lib = jtx._capi.load()
data = jtx.scenes.cornell(); sc = jtx.Scene(data); sc.buildBVH()
cam = data.camera_desc(1920, 1080, 8, 8, 8)
dev = torch.device("cuda", 0)
acc = torch.zeros(1920 * 1080 * 3, dtype=torch.float32, device=dev); img = torch.zeros(1920 * 1080 * 3, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
ms = C.c_float(); nl = C.c_int32()
for n in (64, 16, 8, 1):
    for rep in range(2):
        torch.cuda.synchronize(); lib.jtx_mi_kernel_time(sc.handle, C.byref(ms), C.byref(nl))
        t = time.perf_counter()
        for b in range(0, 64, n):
            jtx.distributed.render_shard(sc, cam, 0, 1, acc, img, stream=st.cuda_stream, sample_begin=b, sample_end=b + n)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t) * 1e3
        lib.jtx_mi_kernel_time(sc.handle, C.byref(ms), C.byref(nl))
    print(f"{n:3d} strata per pass: {nl.value} launches, kernel sum {ms.value:8.2f} ms, wall {wall:8.2f} ms", flush=True)
