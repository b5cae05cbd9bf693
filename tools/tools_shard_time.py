# ON THE GPU BOX: steady-state time per frame of a 1/N tile shard of a bench workload on ONE GPU (rehearsal of the strong-scaling run),
# with one frame at a time (round 4's loop) and with two frames in flight (distributed.ShardPipeline, jtx_mi_render_opts.frame_slot).
# usage: python3 tools/tools_shard_time.py [workload] [frames]
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import jtx_pathtracer_amd as jtx
lib = jtx._capi.load()
wl = sys.argv[1] if len(sys.argv) > 1 else "cornell_1920x1080_64spp_d8"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
name, data, (W, H, xs, ys, depth) = bench.load_workload(jtx, wl)
sc = jtx.Scene(data); sc.buildBVH()
cam = data.camera_desc(W, H, xs, ys, depth)
dev = torch.device("cuda", 0)
print(name)
for world in (1, 2, 4, 8):
    row = []
    for fif in (1, 2, 3):
        pipe = jtx.distributed.ShardPipeline(sc, cam, 0, world, dev, None, integrator=1, frames_in_flight=fif)
        for rep in range(3):
            pipe.step()
        torch.cuda.synchronize()
        ms = C.c_float(); n = C.c_int32(); lib.jtx_mi_kernel_time(sc.handle, C.byref(ms), C.byref(n))
        t0 = time.perf_counter()
        for rep in range(frames):
            pipe.step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / frames * 1e3
        lib.jtx_mi_kernel_time(sc.handle, C.byref(ms), C.byref(n))
        row.append((ms.value / n.value, wall))
        del pipe
    print(f"world {world}: rank-0 shard, ms wall/frame (mean launch duration): " +
          " | ".join(f"{fif} in flight {row[fif - 1][1]:7.3f} ({row[fif - 1][0]:7.3f})" for fif in (1, 2, 3)), flush=True)
# device-side cost of the per-frame exchange (FrameGather pack on every rank, scatter on rank 0), 8-rank geometry
acc = torch.zeros(H * W * 3, dtype=torch.float32, device=dev); img = torch.zeros(H * W * 3, dtype=torch.uint8, device=dev)
fg = jtx.distributed.FrameGather(W, H, 0, 8, dev)
for nm, fn in (("pack", lambda: fg.pack(acc, img)), ("scatter", lambda: fg.scatter(acc, img))):
    for rep in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(20): fn()
    torch.cuda.synchronize()
    print(f"FrameGather.{nm} (world 8): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
