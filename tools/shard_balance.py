# ON THE GPU BOX: how evenly the interleaved 32x32 tiles spread a frame's work over N ranks -- per-rank ray counts (counting pass) and
# steady-state ms per frame of every rank's shard (three frames in flight), one rank after the other on this one GPU.
# usage: python3 tools/shard_balance.py [world] [workload]
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
import jtx_pathtracer_amd as jtx
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = sys.argv[2] if len(sys.argv) > 2 else "cornell_1920x1080_64spp_d8"
lib = jtx._capi.load()
name, data, (W, H, xs, ys, depth) = bench.load_workload(jtx, wl)
sc = jtx.Scene(data); sc.buildBVH()
cam = data.camera_desc(W, H, xs, ys, depth)
dev = torch.device("cuda", 0)
acc = torch.zeros(H * W * 3, dtype=torch.float32, device=dev); img = torch.zeros(H * W * 3, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(device=dev)
rays, ms = [], []
order = [int(x) for x in os.environ.get('RANK_ORDER', '').split(',') if x] or list(range(world))
for r in order:
    jtx.distributed.render_shard(sc, cam, r, world, acc, img, stream=st.cuda_stream, count_rays=True, integrator=1)
    torch.cuda.synchronize()
    c = jtx._capi.Counters(); jtx._capi.check(lib.jtx_mi_get_counters(sc.handle, C.byref(c)))
    rays.append(c.n_closest + c.n_any)
    pipe = jtx.distributed.ShardPipeline(sc, cam, r, world, dev, None, integrator=1)
    if r == order[0]:
        streams = pipe.rstreams                          # every rank's shard on the SAME three streams: which hardware queues a pipeline's
    pipe.rstreams = streams                              # streams land on moves its time by up to 8 % (tools/pipe_repeat.py)
    pipe.prime()
    for _ in range(3):
        pipe.step()
    best = 1e9
    for rep in range(3):                                   # best of three (the first timing of a fresh pipeline comes out up to 8 % slow)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20):
            pipe.step(last=(i == 19))
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    ms.append(best)
    del pipe
tot = sum(rays) * world / len(order)
print(f"{name}, {world} ranks, order {order}: rays per rank / mean = " + " ".join(f"{x * world / tot:.3f}" for x in rays))
print("ms per frame of each rank's shard: " + " ".join(f"{x:.3f}" for x in ms) + f"   max / mean = {max(ms) / (sum(ms) / len(ms)):.3f}, max {max(ms):.3f}, mean {sum(ms) / len(ms):.3f}")
