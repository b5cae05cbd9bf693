#!/usr/bin/env python3
"""ON THE GPU BOX: soak of the create / pin / render / rebuild / destroy path (the intermittent "Memory access fault by
GPU" of round 4 -- three aborts in ~30 suite runs, inside jtx_mi_scene_create, address in the malloc heap -- was mitigated twice, by
library-owned page-locked staging and by film buffers outside the malloc heap, and never seen again; this is the run that bounds it).

Each iteration is what Display::renderScene does after an edit (display.cpp:902-905: rebuild if dirty, then render), from scratch:
    Scene(data) -> buildBVH (host build + upload through the staging buffer) -> StaticCamera with page-locked film buffers -> render
    (uncounted kernel, film delivered to the host) -> film checked against the first iteration's -> transform edit -> rebuildBVH on the
    device -> render -> the frame again with a callback per pass (one progressive launch; every third iteration cancelled from another
    thread) -> [every 5th: four frames in flight on three streams / frame slots] -> destroy; scene, size and strata vary with the iteration, so allocations do not simply recycle.
Every 50th iteration: the two-process rehearsal (two ranks sharing the card: their own contexts, IPC-free gloo exchange).
JTX_ABORT_LOG is armed: a runtime abort leaves its reason and a native backtrace there.

    python3 tools/soak.py [iterations] [seconds] [--no-rehearsal]      -> one summary line; non-zero exit on any mismatch
"""
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
os.environ.setdefault("JTX_ABORT_LOG", os.path.join(ROOT, "gpurun_out", "jtx_abort_soak.log"))
import numpy as np


def rehearsal():
    """two ranks on this card through bench.py's own launcher (gloo, both on device 0), a short frame loop"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(JTX_DIST_BACKEND="gloo", JTX_ALL_RANKS_ON_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--workload", "cornell_512x512_16spp_d4"], capture_output=True, text=True, timeout=300, env=env)
    return r.returncode == 0 and "nranks_seen\": 2" in r.stdout, (r.stdout + r.stderr)[-600:]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    iters = int(args[0]) if args else 300
    budget = float(args[1]) if len(args) > 1 else 1e9
    with_rehearsal = "--no-rehearsal" not in sys.argv
    import torch                                           # (torch brings its own HIP runtime: it initialises first, as in bench.py and conftest.py)
    torch.zeros(1, device="cuda")
    import jtx_pathtracer_amd as jtx
    lib = jtx._capi.load()
    jtx._capi.check(lib.jtx_mi_set_device(0))
    makers = [("cornell", lambda: jtx.scenes.cornell()), ("atrium3k", lambda: jtx.scenes.atrium(target_tris=3000)),
              ("mixed", lambda: jtx.scenes.mixed(sphere_res=(16, 8))), ("atrium20k", lambda: jtx.scenes.atrium(target_tris=20000))]
    sizes = [(96, 64), (200, 136), (321, 177), (64, 64)]
    first = {}
    t0 = time.time()
    done = rehearsals = 0
    for it in range(iters):
        if time.time() - t0 > budget:
            break
        name, make = makers[it % len(makers)]
        W, H = sizes[(it // len(makers)) % len(sizes)]
        xs = 1 + it % 3
        data = make()
        sc = jtx.Scene(data); sc.buildBVH()
        cam = jtx.StaticCamera(W, H, data.camera, xs, 2, 4)                 # film: anonymous mappings, page-locked at the first render
        cam.render(sc, count_rays=False)
        key = (name, W, H, xs)
        crc = zlib.crc32(np.ascontiguousarray(cam.acc_).tobytes()) ^ zlib.crc32(np.ascontiguousarray(cam.img_).tobytes())
        if first.setdefault(key, crc) != crc:
            print(f"MISMATCH at iteration {it}: {key} film differs from its first render"); sys.exit(2)
        # an edit, a device rebuild, another frame (display.cpp:545-588, 902-905)
        m = np.eye(4, dtype=np.float32); m[0, 3] = 1.0 + 0.01 * (it % 7)
        sc.setTransform(0, m)
        sc.rebuildBVHOnDevice(1)
        cam.render(sc, count_rays=False)
        key2 = key + ("edited", it % 7)
        crc = zlib.crc32(np.ascontiguousarray(cam.acc_).tobytes())
        if first.setdefault(key2, crc) != crc:
            print(f"MISMATCH at iteration {it}: {key2} film differs from its first render"); sys.exit(2)
        # round 6: the same frame through a callback per pass -- ONE progressive launch (k_render_paths<PROG> + k_resolve_progressive), the pass
        # size varying -- and, every third iteration, cancelled from another thread at a varying moment: the film must then be exactly the
        # strata [0, currentSample_) (a render of that range, bit for bit)
        prog = jtx.StaticCamera(W, H, data.camera, xs, 2, 4); prog.samplesPerPass_ = 1 + it % 2
        seen = []
        prog.render(sc, count_rays=False, progress=lambda c, t: seen.append(c))
        if zlib.crc32(np.ascontiguousarray(prog.acc_).tobytes()) != crc or seen[-1] != 2 * xs:
            print(f"MISMATCH at iteration {it}: {key2} progressive film differs"); sys.exit(2)
        if it % 3 == 2:
            import threading
            timer = threading.Timer(0.0002 * (1 + it % 11), prog.terminateRender)
            timer.start(); prog.render(sc, count_rays=False, progress=lambda c, t: None); timer.join()
            n = prog.currentSample_
            part = jtx.StaticCamera(W, H, data.camera, xs, 2, 4)
            if n:
                part.render(sc, count_rays=False, sample_begin=0, sample_end=n)
            if not (np.array_equal(prog.acc_.view(np.uint32), part.acc_.view(np.uint32)) and np.array_equal(prog.img_, part.img_)):
                print(f"MISMATCH at iteration {it}: {key2} cancelled progressive film is not the strata [0, {n})"); sys.exit(2)
            part._unpin(); del part
        prog._unpin(); del prog
        if it % 5 == 4:
            # ... and four frames in flight on three streams / frame slots, the scene destroyed right behind them
            dev = torch.device("cuda", 0)
            pipe = jtx.distributed.ShardPipeline(sc, data.camera_desc(W, H, xs, 2, 4), 0, 1, dev, None, integrator=1)
            for k in range(4):
                pipe.step(last=(k == 3))
            torch.cuda.synchronize()
            got = pipe.accs[0].cpu().numpy()
            if zlib.crc32(np.ascontiguousarray(got).tobytes()) != crc:
                print(f"MISMATCH at iteration {it}: {key2} pipelined film differs"); sys.exit(2)
            del pipe
        cam._unpin()
        sc.destroy()
        del cam, sc, data
        done += 1
        if with_rehearsal and it % 50 == 49:
            ok, tail = rehearsal()
            rehearsals += 1
            if not ok:
                print(f"REHEARSAL FAILED at iteration {it}:\n{tail}"); sys.exit(3)
        if it % 25 == 24:
            print(f"[soak] {it + 1} iterations, {time.time() - t0:.0f} s", flush=True)
    log = os.environ["JTX_ABORT_LOG"]
    aborted = os.path.exists(log) and os.path.getsize(log) > 0
    print(f"soak: {done} iterations ({len(first)} distinct frames, each re-rendered bit-identically), {rehearsals} two-process rehearsals, "
          f"{time.time() - t0:.0f} s, 0 faults, abort log {'NOT EMPTY' if aborted else 'empty'}")
    sys.exit(4 if aborted else 0)


if __name__ == "__main__":
    main()
