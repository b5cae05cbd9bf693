# diagnostic: wave start / end spread of a 1/N shard of the C2 frame (build with -DJTX_PROFILE_TIMELINE); usage: shard_timeline.py [world]
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jtx_pathtracer_amd as jtx
lib = jtx._capi.load()
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
data = jtx.scenes.cornell(); sc = jtx.Scene(data); sc.buildBVH()
W, H = 1920, 1080
cam = data.camera_desc(W, H, 8, 8, 8)
dev = torch.device("cuda", 0)
acc = torch.zeros(H * W * 3, dtype=torch.float32, device=dev); img = torch.zeros(H * W * 3, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for rep in range(3):
    jtx.distributed.render_shard(sc, cam, 0, world, acc, img, stream=st.cuda_stream)
torch.cuda.synchronize()
n = 7168
buf = (C.c_uint64 * (2 * n))()
f = lib.jtx_mi_debug_timeline; f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
assert f(sc.handle, buf, n) == 0
t = np.array(buf, dtype=np.int64).reshape(n, 2).astype(np.float64)
t = t[t[:, 1] > 0]
t0 = t[:, 0].min(); t -= t0
tick = 1e-5
s, e = t[:, 0] * tick, t[:, 1] * tick
print(f"world {world}: {len(t)} waves, launch {e.max():.3f} ms; starts p50 {np.percentile(s, 50):.3f} p99 {np.percentile(s, 99):.3f} max {s.max():.3f}; "
      f"ends p1 {np.percentile(e, 1):.3f} p10 {np.percentile(e, 10):.3f} p50 {np.percentile(e, 50):.3f} p90 {np.percentile(e, 90):.3f} max {e.max():.3f}")
print(f"wave-ms of work {(e - s).sum():.1f} = {(e - s).sum() / len(t):.3f} ms per wave; efficiency {(e - s).sum() / len(t) / e.max():.3f}")
