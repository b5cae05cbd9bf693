"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of the pixel-tile sharding + per-frame reduce.

The GPU renderer cannot run here, so each rank fills its shard from the CPU oracle's frame masked by
the product's ownership rule (jtx.distributed.tile_owner_mask) -- exactly what jtx_mi_render_device
produces for tile_rank/tile_world (the GPU test test_render_tile_sharding checks that) -- and the
product's reduce_frame() sums the shards over torch.distributed.  Rank 0 must end up with the 1-rank
frame bit for bit.
"""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, w, h, out_dir, mode="reduce"):
    import oracle_lib as ol
    import jtx_pathtracer_amd as jtx
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = jtx.scenes.cornell()
        cam = data.camera_desc(w, h, 2, 1, 3)
        acc, img, _ = ol.OracleScene(data).render(cam, threads=2)
        mask = jtx.distributed.tile_owner_mask(w, h, rank, world)
        assert jtx.distributed.owned_tiles(w, h, rank, world) == len(
            {(r // 32, c // 32) for r, c in zip(*np.nonzero(mask))})
        my_acc = torch.from_numpy(np.where(mask[..., None], acc, 0).astype(np.float32)).reshape(-1).clone()
        my_img = torch.from_numpy(np.where(mask[..., None], img, 0).astype(np.uint8)).reshape(-1).clone()
        if mode == "reduce":
            jtx.distributed.reduce_frame(my_acc, my_img, dst=0)
        else:
            # what is outside the own tiles must not matter to the gather: poison it
            my_acc[torch.from_numpy(~np.repeat(mask.reshape(-1), 3))] = float("nan")
            my_img[torch.from_numpy(~np.repeat(mask.reshape(-1), 3))] = 77
            fg = jtx.distributed.FrameGather(w, h, rank, world, torch.device("cpu"))
            fg.collect(my_acc, my_img)
            fg.collect(my_acc, my_img)          # idempotent on the root, frame after frame
        if rank == 0:
            ok = np.array_equal(my_acc.numpy().view(np.uint32), acc.reshape(-1).view(np.uint32)) and \
                np.array_equal(my_img.numpy(), img.reshape(-1))
            open(os.path.join(out_dir, "result"), "w").write("ok" if ok else "mismatch")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 100, 70), (3, 65, 33)])
def test_tile_shard_reduce_gloo(world, w, h, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, w, h, str(tmp_path)), nprocs=world, join=True)
    assert open(tmp_path / "result").read() == "ok"


@pytest.mark.parametrize("world,w,h", [(2, 100, 70), (3, 65, 33), (3, 40, 20)])
def test_tile_shard_gather_gloo(world, w, h, tmp_path):
    """FrameGather (the default per-frame exchange): compact own-pixel slabs gathered to rank 0; (3, 40, 20) has
    only two tiles, so one rank owns nothing."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, w, h, str(tmp_path), "gather"), nprocs=world, join=True)
    assert open(tmp_path / "result").read() == "ok"


def test_owner_masks_partition_the_frame():
    import jtx_pathtracer_amd as jtx
    for (w, h, world) in [(1920, 1080, 8), (33, 7, 2), (1, 1, 4), (64, 64, 3)]:
        cover = sum(jtx.distributed.tile_owner_mask(w, h, r, world).astype(np.int32) for r in range(world))
        assert (cover == 1).all()
        tiles = ((w + 31) // 32) * ((h + 31) // 32)
        assert sum(jtx.distributed.owned_tiles(w, h, r, world) for r in range(world)) == tiles


def test_tile_slot_maps_cpp(tmp_path):
    """The C++ side of the exchange (csrc/jtx_tiles.hpp, used by jtx_multi.hip's pack / scatter kernels and by the render
    kernels): slot <-> pixel maps are inverse bijections for ragged sizes and 1..8 shards.  Pure host C++, no GPU."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "tile_map_test")
    subprocess.run(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(root, "tests", "cpp", "tile_map_test.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "tile maps ok" in r.stdout, r.stdout + r.stderr


def test_python_and_cpp_tile_maps_agree():
    """distributed.tile_owner_mask (what the torchrun path packs / scatters by) and jtx_tiles.hpp's pixelToSlot name the
    same owner for every pixel."""
    import numpy as np
    from jtx_pathtracer_amd import distributed as D
    for (w, h, world) in [(200, 100, 3), (1920, 1080, 8), (33, 65, 2)]:
        tiles_x = (w + 31) // 32
        rows, cols = np.mgrid[0:h, 0:w]
        owner = ((rows // 32) * tiles_x + cols // 32) % world           # pixelToSlot's rank
        for r in range(world):
            assert np.array_equal(D.tile_owner_mask(w, h, r, world), owner == r)


def test_bench_self_launch_relays_the_ranks_return_code():
    """`python bench.py --gpus 2` without a launcher starts torch.distributed.run itself (as a child process of a parent that has not
    imported torch).  Without a GPU the two ranks stop with bench.py's "needs an MI355X" (there is no CPU fallback): the parent relays
    that failure as ITS return code, and the message shows that both ranks got as far as bench.py's main().  Round 6:
    a failed launch leaves ONE JSON line -- the error, how many ranks were seen and how far each came -- so that a driver-run record of a
    failed multi-GPU run diagnoses itself."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""                          # (were this ever run on a GPU box: still the no-device branch)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr and "launch N>1 with" not in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    rec = json.loads(lines[0])
    assert rec["value"] is None and rec["n_gpus"] == 2 and rec["returncode"] == r.returncode and "exited with code" in rec["error"]
    assert rec["nranks_seen"] == 2 and [x["rank"] for x in rec["ranks"]] == [0, 1]
    assert all(x["stage"] == "started" and x["devices_visible"] == 0 for x in rec["ranks"])      # both ranks came up; neither found a device
    assert "needs an MI355X" in rec["stderr_tail"]
