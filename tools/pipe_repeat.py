import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("Q", "8"))
import torch, bench
import jtx_pathtracer_amd as jtx
name, data, (W, H, xs, ys, depth) = bench.load_workload(jtx, "cornell_1920x1080_64spp_d8")
sc = jtx.Scene(data); sc.buildBVH()
cam = data.camera_desc(W, H, xs, ys, depth)
dev = torch.device("cuda", 0)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
out = []
for k in range(12):
    pipe = jtx.distributed.ShardPipeline(sc, cam, 0, world, dev, None, integrator=1)
    if os.environ.get("PRIO"):
        lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
        pr = [int(x) for x in os.environ["PRIO"].split(",")]
        pipe.rstreams = [torch.cuda.Stream(device=dev, priority=pr[k % len(pr)]) for k in range(len(pipe.rstreams))]
    pipe.prime()
    for _ in range(3): pipe.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): pipe.step(last=(i == 19))
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 20 * 1e3)
    ids = [s.cuda_stream & 0xffff for s in pipe.rstreams]
    del pipe
print("priorities", os.environ.get("PRIO"), end=" ")
print(f"world {world} queues {os.environ['GPU_MAX_HW_QUEUES']}: " + " ".join(f"{x:.3f}" for x in out))
