#!/usr/bin/env python3
"""RCCL on the box at hand, through the calls the N > 1 bench path makes (bench.py: init_process_group("nccl", device_id=...);
distributed.FrameGather: pack -> dist.gather of uint8 slabs -> scatter; distributed.reduce_frame: dist.reduce of the f32 film).

A one-GPU box has one rank to offer, so the group has ONE member: the collectives degenerate to self-copies, but they go through
RCCL's communicator set-up, its stream handling and torch's c10d argument checks for exactly the tensor shapes / dtypes / views
of the product path -- the part of the multi-GPU path that gloo rehearsals cannot touch.  Prints one JSON line.

    python tools/rccl_selfcheck.py [--width 1920 --height 1080]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--port", type=int, default=29533)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import jtx_pathtracer_amd as jtx
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(args.port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    t0 = time.time()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    W, H = args.width, args.height
    n = W * H
    g = torch.Generator(device="cpu").manual_seed(7)
    acc = torch.rand(n * 3, generator=g).to(dev)
    img = torch.randint(0, 256, (n * 3,), generator=g, dtype=torch.uint8).to(dev)
    fg = jtx.distributed.FrameGather(W, H, 0, 1, dev)
    out_acc, out_img = torch.zeros_like(acc), torch.zeros_like(img)
    # FrameGather.collect returns early for a group of one: the same three steps, spelled out
    fg.pack(acc, img)
    dist.gather(fg.slab, list(fg.recv.unbind(0)), dst=0)
    fg.scatter(out_acc, out_img)
    torch.cuda.synchronize()
    ok_gather = bool(torch.equal(out_acc.view(torch.int32), acc.view(torch.int32)) and torch.equal(out_img, img))
    # the alternative collective: one sum-reduce of the full-size buffers
    a2, i2 = acc.clone(), img.clone()
    dist.reduce(a2, dst=0, op=dist.ReduceOp.SUM)
    dist.reduce(i2, dst=0, op=dist.ReduceOp.SUM)
    # the rows bench.py gathers at the end (per-rank timings) and its barrier
    row = torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=torch.float64, device=dev)
    rows = [torch.zeros_like(row)]
    dist.all_gather(rows, row)
    dist.barrier()
    torch.cuda.synchronize()
    ok_reduce = bool(torch.equal(a2.view(torch.int32), acc.view(torch.int32)) and torch.equal(i2, img) and torch.equal(rows[0], row))
    dist.destroy_process_group()
    print(json.dumps({"rccl_selfcheck": "ok" if ok_gather and ok_reduce else "MISMATCH", "gather": ok_gather, "reduce": ok_reduce,
                      "slab_bytes": int(fg.slab.numel()), "seconds": round(time.time() - t0, 2),
                      "nccl_version": ".".join(map(str, torch.cuda.nccl.version()))}))
    return 0 if ok_gather and ok_reduce else 1


if __name__ == "__main__":
    sys.exit(main())
