# diagnostic: rank-0 shard kernel time of the C2 frame for world x strata-groups (JTX_STRATA_GROUPS is read per launch)
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
ms = C.c_float(); n = C.c_int32()
for world in (1, 2, 4, 8):
    row = []
    for g in (1, 2, 4, 8, 16, 32, 64):
        os.environ["JTX_STRATA_GROUPS"] = str(g)
        jtx.distributed.render_shard(sc, cam, 0, world, acc, img, stream=st.cuda_stream)
        torch.cuda.synchronize(); lib.jtx_mi_kernel_time(sc.handle, C.byref(ms), C.byref(n))
        for rep in range(3):
            jtx.distributed.render_shard(sc, cam, 0, world, acc, img, stream=st.cuda_stream)
        torch.cuda.synchronize(); lib.jtx_mi_kernel_time(sc.handle, C.byref(ms), C.byref(n))
        row.append(ms.value / n.value)
    print(f"world {world}: groups 1,2,4,8,16,32,64 ->", " ".join(f"{x:7.2f}" for x in row), "  ideal %.2f" % (0 if world == 1 else 0), flush=True)
