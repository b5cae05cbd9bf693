#!/usr/bin/env python3
"""bench.py -- Mrays/s of the JTX path-tracing hot path on MI355X (BASELINE.json metric).

A "step" is one whole frame of the workload: BASELINE config 2, Cornell box 1920x1080, 64 spp (8x8
strata), maxDepth 8, through jtx_mi_render_device (scene already resident in HBM), three frames in flight,
EVERY FILM DELIVERED TO THE HOST (acc + img in page-locked memory: SURVEY 8d's "last byte on the host"; round 6).
With N > 1 ranks the frame's 32x32 pixel tiles are interleaved over the ranks and one RCCL exchange per frame
brings the disjoint shards to rank 0, which delivers the assembled frame (strong scaling: the frame is fixed).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          (no launcher: bench.py starts that torch.distributed.run itself, as a child process)

Prints ONE JSON line on rank 0.  ray = one Scene::closestHit or Scene::anyHit call.

N = 1: after the headline's timed region the other BASELINE.json workloads (C3 atrium, C5 mixed, C1) are timed for a few frames
each and ride in the same line under "workloads" (outside `value` / `steps` / `ms_per_step`, which stay the headline's);
`--headline-only` skips them.  `--scene <file.glb|.gltf|.obj>` times an asset through the loaders (createScene, scene.cpp:176-209)
instead of a procedural scene; a `sponza.gltf` / `sponza.glb` dropped into `assets/` replaces the procedural atrium.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene factory name, width, height, xs, ys, max_depth)
    "cornell_1920x1080_64spp_d8": ("cornell", 1920, 1080, 8, 8, 8),
    "cornell_512x512_16spp_d4": ("cornell", 512, 512, 4, 4, 4),
    "atrium_1920x1080_64spp_d8": ("atrium", 1920, 1080, 8, 8, 8),
    "mixed_1920x1080_128spp_d8": ("mixed", 1920, 1080, 16, 8, 8),
    # BASELINE config 4 (8 GPUs: pixel-tile shard + one exchange per frame): `--workload atrium_3840x2160_256spp_d8 --gpus 8`; on one GPU
    # the frame's 34 GB of radiance records go in passes under the 8 GiB cap (~5 s per frame).  Not part of the default run.
    "atrium_3840x2160_256spp_d8": ("atrium", 3840, 2160, 16, 16, 8),
}
DEFAULT_EXTRAS = ("cornell_512x512_16spp_d4", "atrium_1920x1080_64spp_d8", "mixed_1920x1080_128spp_d8")


EXTRA_STEPS = {"atrium_1920x1080_64spp_d8": 3, "mixed_1920x1080_128spp_d8": 3, "cornell_512x512_16spp_d4": 20}


def sponza_asset():
    """"if a real sponza.gltf is dropped in assets/ the bench uses it instead" (of the procedural atrium)."""
    for n in ("sponza.gltf", "sponza.glb", "Sponza.gltf", "Sponza.glb"):
        p = os.path.join(ROOT, "assets", n)
        if os.path.exists(p):
            return p
    return None


def frame_camera_inside(data):
    """A file scene comes with createScene's camera (0,0,8) -> origin, yfov 20 (scene.cpp:196-201), which looks AT an object.  For an
    interior (Sponza) the bench puts the camera inside instead, deterministically from the bounds: a quarter along the longest
    horizontal axis, 30 % up, looking down that axis, yfov 60."""
    import numpy as np
    lo = np.min([np.min(m["vertices"] if m.get("transform") is None else m["vertices"] @ np.asarray(m["transform"], np.float32)[:3, :3].T
                        + np.asarray(m["transform"], np.float32)[:3, 3], axis=0) for m in data.meshes], axis=0)
    hi = np.max([np.max(m["vertices"] if m.get("transform") is None else m["vertices"] @ np.asarray(m["transform"], np.float32)[:3, :3].T
                        + np.asarray(m["transform"], np.float32)[:3, 3], axis=0) for m in data.meshes], axis=0)
    ax = 0 if (hi[0] - lo[0]) >= (hi[2] - lo[2]) else 2
    c = 0.5 * (lo + hi)
    c[1] = lo[1] + 0.3 * (hi[1] - lo[1])
    t = c.copy()
    c[ax] = lo[ax] + 0.25 * (hi[ax] - lo[ax]); t[ax] = hi[ax]
    data.camera = dict(center=tuple(float(x) for x in c), target=tuple(float(x) for x in t), up=(0, 1, 0), yfov=60.0,
                       defocus_angle=0.0, focus_distance=1.0)


def load_workload(jtx, name, atrium_tris=262144, scene_file=None, camera=None):
    """-> (workload name, SceneData, (W, H, xs, ys, depth)).  `scene_file`: an asset through the reference's createScene rules
    (scenes.create_scene: OBJ / glTF / GLB loaders, loader.cpp:13-225, scene.cpp:176-209), named after the file."""
    factory, W, H, xs, ys, depth = WORKLOADS[name]
    if scene_file is None and factory == "atrium":
        sp = sponza_asset()
        if sp is not None:
            scene_file, camera = sp, camera or "inside"
    if scene_file is not None:
        data = jtx.scenes.create_scene(scene_file)
        if camera == "inside":
            frame_camera_inside(data)
        elif camera:
            v = [float(x) for x in camera.split(",")]
            if len(v) != 7:
                raise SystemExit("--camera cx,cy,cz,tx,ty,tz,yfov")
            data.camera.update(center=tuple(v[0:3]), target=tuple(v[3:6]), yfov=v[6])
        base = os.path.splitext(os.path.basename(scene_file))[0]
        return f"file:{base}_{W}x{H}_{xs * ys}spp_d{depth}", data, (W, H, xs, ys, depth)
    data = getattr(jtx.scenes, factory)(atrium_tris) if factory == "atrium" else getattr(jtx.scenes, factory)()
    return name, data, (W, H, xs, ys, depth)


def time_workload(jtx, torch, dev, tstream, name, data, dims, steps, warmup, integrator=None):
    """One rank, one workload: counted pass (device counters), warm-up, `steps` timed frames through jtx_mi_render_device
    (film resident in HBM, resolve pass included), live HIP-event kernel time -> the dict that rides under "workloads"."""
    lib = jtx._capi.load()
    W, H, xs, ys, depth = dims
    scene = jtx.Scene(data)
    scene.buildBVH()
    try:
        cam = data.camera_desc(W, H, xs, ys, depth)
        acc = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
        img = torch.zeros(H * W * 3, dtype=torch.uint8, device=dev)
        if integrator is None:
            integrator = scene.info()["auto_integrator"]
        pipe = jtx.distributed.ShardPipeline(scene, cam, 0, 1, dev, None, integrator=integrator, frames_in_flight=frames_in_flight(),
                                             deliver_to_host=True)

        def frame(count=False, last=False):
            if count:
                jtx.distributed.render_shard(scene, cam, 0, 1, acc, img, stream=tstream.cuda_stream, count_rays=True, integrator=integrator)
            else:
                pipe.step(last=last)
        frame(count=True)
        torch.cuda.synchronize()
        cnt = jtx._capi.Counters()
        jtx._capi.check(lib.jtx_mi_get_counters(scene.handle, C.byref(cnt)))
        mine = cnt.as_dict()
        rays = mine["n_closest"] + mine["n_any"]
        pipe.prime()
        for _ in range(warmup):
            frame()
        torch.cuda.synchronize()
        ms = C.c_float(); nl = C.c_int32()
        jtx._capi.check(lib.jtx_mi_kernel_time(scene.handle, C.byref(ms), C.byref(nl)))
        t = time.perf_counter()
        for i in range(steps):
            frame(last=(i == steps - 1))
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t
        kernel_ms = serial_kernel_ms(jtx, torch, lib, scene, cam, 0, 1, dev, integrator, frames=min(3, steps))
        info = scene.info()
        out = {"ms_per_step": round(elapsed / steps * 1e3, 3), "kernel_ms": round(kernel_ms, 4), "steps": steps, "warmup": warmup,
               "frames_in_flight": len(pipe.rstreams), "timed_region": "film delivered to host",
               "value": round(rays * steps / elapsed / 1e6, 2), "unit": "Mrays/s", "rays_per_frame": rays,
               "scene_triangles": data.num_triangles, "integrator": integrator}
        if integrator == 1:
            sinfo = {"lds_resident": bool(info["lds_resident"]), "lds_bytes": 8 * 32 * info["num_nodes"] + 48 * info["num_prims"],
                     "workgroups": info["resident_workgroups"]}
            roof = roofline_block(name, sinfo, mine, "k_render_paths", kernel_ms, 1, info["num_cus"])
            if len(pipe.rstreams) > 1:
                out["in_flight"] = in_flight_block(roof, len(pipe.rstreams), elapsed / steps * 1e3, None)
            for k in ("useful_frac", "frac", "traffic", "vector_memory", "lane_util", "issue_model", "pmc_stale"):
                if k in roof:
                    out[k] = roof[k] if k != "issue_model" else {"busy": roof[k]["busy"], "busy_calibrated": roof[k]["busy_calibrated"], "co_issue": roof[k]["co_issue"]}
            out["hbm_measured_frac"] = roof["hbm"].get("measured_frac")
            out["useful_lane_ops_per_launch"] = roof["useful"]["lane_ops_per_launch"]; out["num_cus"] = roof["num_cus"]
            out["kernel"] = roof["kernel"]
        return out
    finally:
        scene.destroy()


def in_flight_block(roof, n, ms_per_frame, kernel_ms_timed):
    """the fractions of `roof` (measured on launches with one frame in flight) restated on the timed region's steady state: the chip
    runs nothing but these frames, so a frame's share of it is the wall time per frame (the resolve pass included: a lower bound)"""
    t = ms_per_frame * 1e-3
    b = {"frames_in_flight": n, "ms_per_frame": round(ms_per_frame, 3),
         "useful_frac": round(roof["useful"]["lane_ops_per_launch"] / t / 1e12 / roof["useful"]["peak"], 4),
         "note": "the same numerators over the wall time per frame of the timed region (frames of the scene overlap: the last chunks of a "
                 "launch run beside the first chunks of the next, the resolve pass beside the one after; roofline.kernel_ms / frac / "
                 "useful_frac are measured on launches with ONE frame in flight, as JTX_FRAMES_IN_FLIGHT=1 times them)"}
    if roof.get("valu_instructions_per_launch"):
        b["frac"] = round(roof["valu_instructions_per_launch"] / t / 1e9 / roof["peak"], 4)
    if kernel_ms_timed is not None:
        b["launch_ms_start_to_end"] = round(kernel_ms_timed, 3)
    return b


def frames_in_flight():
    """frames of one scene in flight in the timed loops (distributed.ShardPipeline; JTX_FRAMES_IN_FLIGHT=1: round 4's one-stream loop)"""
    return max(1, min(3, int(os.environ.get("JTX_FRAMES_IN_FLIGHT", "3"))))


def serial_kernel_ms(jtx, torch, lib, scene, cam, rank, world, dev, integrator, frames=3):
    """average duration of the dominant kernel's launch with ONE frame in flight (what `JTX_FRAMES_IN_FLIGHT=1 bench.py` times, and what
    a rocprofv3 kernel trace of that command shows): the roofline's denominator.  In the timed region several frames of the scene are in
    flight and consecutive launches share the chip -- a launch then lasts longer than a frame takes (its own start-to-end time counts
    the lanes it left to its neighbours), so its duration says nothing about the chip time it used."""
    pipe = jtx.distributed.ShardPipeline(scene, cam, rank, world, dev, None, integrator=integrator, frames_in_flight=1)
    ms = C.c_float(); nl = C.c_int32()
    pipe.step()
    torch.cuda.synchronize()
    jtx._capi.check(lib.jtx_mi_kernel_time(scene.handle, C.byref(ms), C.byref(nl)))     # drop what came before
    for _ in range(max(1, frames)):
        pipe.step()
    torch.cuda.synchronize()
    jtx._capi.check(lib.jtx_mi_kernel_time(scene.handle, C.byref(ms), C.byref(nl)))
    return ms.value / max(1, nl.value)


def algorithmic_bytes(c):
    """bytes the reference's data structures imply per ray / shade / sample."""
    b_closest = 32 * c["n_nodes_closest"] + 56 * c["n_tri_closest"] + 60 * c["n_accept"] + 64 * c["n_closest"]
    b_any = 32 * c["n_nodes_any"] + 56 * c["n_tri_any"] + 64 * c["n_any"]
    b_shade = 176 * c["n_shade"]
    b_cam = 27 * c["n_camera"]
    return b_closest + b_any + b_shade + b_cam


# Useful work of the REFERENCE's algorithm, per counted event, in fp32 lane-operations (one add / sub / mul / div / min / max /
# compare / sqrt = 1; abs / negate are operand modifiers = 0; integer RNG / hashing and address arithmetic = 0; derivation from the
# reference's source in DESIGN.md section 6).  Minimal forms: 1/d once per ray, baked triangle edges, one shading-normal interpolation
# per path vertex -- what any implementation of the same mathematics must do, NOT what this kernel issues: phase A's box tests of
# leaves the reference never visits, scheduling ballots, spill code and idle lanes all LOWER the fraction built on it.
# Every entry is (fast, slow, transcendental): how many of the operations are add / sub / mul (2.4 issue cycles per wave64 instruction
# on gfx950, profiles/r03_valu_rates.txt), min / max / compare / convert (4.4) and divide / sqrt (8.4, priced as ONE reciprocal-class
# instruction each) -- the split that `useful_frac_attainable` prices.
USEFUL_MIX = {
    "ray": (0, 3, 3),        # 3 divisions 1/d + 3 sign tests (scene.cpp:11-12), per closestHit / anyHit call
    "node": (12, 13, 0),     # AABB::hit aabb.hpp:66-81: per axis 2 sub, 2 mul (12), min, max (6); t0 = max of 4, t1 = min of 4 (6); t0 <= t1 (1)
    "tri": (45, 9, 1),       # Moeller-Trumbore run to the end, mesh.hpp:106-127 / 168-192: cross 9, dot 5, |det| test 1, 1/det 1, tvec 3,
                             #   b1 6 + 2 tests, cross 9, b2 6 + add + 2 tests, t 6 + 2 interval tests (most tests leave earlier: upper bound per test)
    "camera": (34, 4, 2),    # Camera::getRay camera.hpp:127-139 (stratum offsets 8, viewport point 14, direction 3, rng floats 6) + clamp
                             #   and accumulate camera.cpp:110-112 / image.hpp:82-86 (9)
    # one vertex of integrateMIS (integrator.cpp:171-216), whatever its material: hit point + normal / uv interpolation + face-forward 42
    # (mesh.hpp:129-145), Light::sample 26 + shadow-ray set-up 16 + rng floats 10 (integrator.cpp:134-150), Frame::fromZ + toLocal 31 and
    # toWorld 15 (bxdf.cpp:10-12, 75), beta update 15, radiance add 6, next ray 6, wo 3
    "vertex": (142, 10, 18),
    # ... and what sampleLights adds when the light is NOT occluded, whatever the material (integrator.cpp:151-166): evalBxdf's and pdfBxdf's
    # own Frame::fromZ + 2 toLocal each (92, bxdf.cpp:80-82, 131-133), f * absdot 9, pl 2, powerHeuristic 6, misWeight * f * L / pl 9, beta * .. + 6
    "unoccluded": (112, 5, 7),
}
# The BxDF's own share of a vertex, by BxDF class (jtx_mi_counters::n_shade_class): its `sample` (sampleBxdf, bxdf.cpp:9-77) ...
#   0 DIFFUSE diffuse.hpp:15-22: cosine-hemisphere sample incl. sin / cos 60, pdf / f / checks 12
#   1 DIELECTRIC rough dielectric.hpp:72-108: sampleWm 130 (microfacet.hpp:86-106), Fresnel 29, then reflection (reflect 12, GGX pdf 91, D G 112:
#     383) or transmission (refract 26, dn 20, pdf 85, f 118: 419) -- the mean
#   2 CONDUCTOR rough conductor.hpp:40-59: sampleWm 130, reflect 12, GGX pdf 90, the complex Fresnel term per channel 3 x 88 + 6, D G / (4 ci co) 116
#   3 METALLIC_ROUGHNESS gltf.hpp:36-90: two lerps 20, specular probability 38, then the GGX lobe (sampleWm 130, reflect 15, pdf 90: 240) or the
#     cosine lobe (82); Schlick 19, diffuse + specular terms 130 -- the mean of the two lobes
#   4 ThinDielectric dielectric.hpp:170-200: Fresnel 24, the inter-reflection term 8, one lobe 9
#   5 DIELECTRIC smooth dielectric.hpp:44-70: Fresnel 24, probabilities 3, reflection 4 or refraction 29 -- the mean
#   6 CONDUCTOR smooth conductor.hpp:33-38: mirror direction, 3 x 88 Fresnel, / cos
USEFUL_SAMPLE_CLASS = [72, 400, 627, 371, 41, 45, 272, 0]
# ... and its `evaluate` + `pdf` when the light sample is not occluded (evalBxdf bxdf.cpp:79-128 + pdfBxdf :130-166):
#   0: same-hemisphere test, R / pi, cos / pi; 1 dielectric.hpp:110-161: half vector + checks 46, Fresnel 29, D G 111; again 46 + 29 for the pdf + GGX pdf 93;
#   2 conductor.hpp:8-29, 61-73: half vector 22, Fresnel 270, D G 116; pdf 114; 3 gltf.hpp:14-34, 92-122: eval 191, pdf 158; 4: evaluate = {}, pdf = 0;
#   5, 6: the smooth tests only
USEFUL_EVAL_CLASS = [5, 356, 524, 349, 2, 6, 4, 0]
CLASS_MIX = {0: (0.75, 0.18, 0.07)}       # share of (fast, slow, transcendental) operations in a BxDF's own code: Lambert as counted above ...
GGX_MIX = (0.78, 0.12, 0.10)              # ... the GGX / Fresnel code is richer in divisions and square roots
ISSUE_CYCLES = (2.4, 4.4, 8.4)            # per wave64 instruction and SIMD (profiles/r03_valu_rates.txt)
USEFUL_OPS = {k: sum(v) for k, v in USEFUL_MIX.items()}
USEFUL_OPS["shade"] = USEFUL_OPS["vertex"] + USEFUL_SAMPLE_CLASS[0]      # (= 242: one occluded Lambert vertex, rounds 2-4's price of EVERY vertex)


def useful_events(c):
    """[(events, (fast, slow, transcendental) operations per event)] of one frame's counters"""
    ev = [(c["n_closest"] + c["n_any"], USEFUL_MIX["ray"]), (c["n_nodes_closest"] + c["n_nodes_any"], USEFUL_MIX["node"]),
          (c["n_tri_closest"] + c["n_tri_any"], USEFUL_MIX["tri"]), (c["n_camera"], USEFUL_MIX["camera"]), (c["n_shade"], USEFUL_MIX["vertex"])]
    sc, ec = c.get("n_shade_class"), c.get("n_eval_class")
    if sc is None:                     # counters of rounds 1-4 (no per-class tallies): every vertex priced as an occluded Lambert one
        sc, ec = [c["n_shade"]] + [0] * 7, [0] * 8
    ev.append((sum(ec), USEFUL_MIX["unoccluded"]))
    for k in range(8):
        mix = CLASS_MIX.get(k, GGX_MIX)
        for n, ops in ((sc[k], USEFUL_SAMPLE_CLASS[k]), (ec[k], USEFUL_EVAL_CLASS[k])):
            if n and ops:
                ev.append((n, tuple(ops * m for m in mix)))
    return ev


def useful_lane_ops(c):
    return sum(n * sum(mix) for n, mix in useful_events(c))


def useful_issue_cycles(c):
    """SIMD cycles the same operations take at the least when every instruction carries 64 lanes of them and issues at its class's rate"""
    return sum(n * sum(m * cy for m, cy in zip(mix, ISSUE_CYCLES)) for n, mix in useful_events(c)) / 64.0


def flat_counters(c):
    """{"n_camera": .., "n_shade_class": [..]} -> a flat {name: int} (for the all-reduce over ranks) and back"""
    out = {}
    for k, v in c.items():
        if isinstance(v, (list, tuple)):
            for i, x in enumerate(v):
                out[f"{k}#{i}"] = int(x)
        else:
            out[k] = int(v)
    return out


def unflat_counters(f):
    out = {}
    for k, v in f.items():
        if "#" in k:
            name, i = k.split("#")
            out.setdefault(name, [])
            while len(out[name]) <= int(i):
                out[name].append(0)
            out[name][int(i)] = v
        else:
            out[k] = v
    return out


def pmc_file():
    """the newest profiles/rNN_pmc.json (rocprofv3 --pmc passes over the timed launches, tools/pmc_collect.sh + pmc_merge.py)"""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")))
    return cands[-1] if cands else None


def source_hash():
    """sha1 over the kernel sources: ties a profiles/ counter file to the build it was collected from."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "jtx-pathtracer_amd", "csrc")
    # what the timed kernel (k_render_paths + k_resolve_samples) is compiled from; the wide-node BUILDER lives in
    # jtx_capi.hip and shows in the counter file's wide_stats (node steps per ray) instead
    for f in ("jtx_kernels.hip", "jtx_scene_dev.hpp", "jtx_wide_quant.hpp", "jtx_bxdf.hpp", "jtx_device_math.hpp", "jtx_launch.hpp", "jtx_tiles.hpp",
              "jtx_profile.hpp"):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    # ... and the flags they are compiled with (round 5: -fno-slp-vectorize changed every kernel's code without touching a source line)
    bp = open(os.path.join(ROOT, "jtx-pathtracer_amd", "build.py")).read()
    h.update(bp[bp.index("FLAGS = ["):bp.index("]", bp.index("FLAGS = [")) + 1].encode())
    return h.hexdigest()[:16]


def issue_calibration(lds_resident):
    """measured / priced cycles of the kernels' own loop bodies replayed at saturation (tools/isa_replay.py on the GPU box ->
    profiles/rNN_issue_replay.txt): the flat-leaf-list kernels (LDS-resident scenes) are calibrated on phase A, the 8-ary
    traversal on the mean of its closestHit and anyHit node steps.  None: no record."""
    import glob, re
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_issue_replay.txt")))
    if not cands:
        return None
    ratio = {}
    for line in open(cands[-1]):
        m = re.match(r"(\w+): (\d+) VALU instructions.*measured / model = ([0-9.]+)", line)
        if m:
            ratio[m.group(1)] = (float(m.group(3)), int(m.group(2)))
    # (leaf list: ONE box of phase A as the compiler emits it, repeated -- tools/micro/rate10; the replay of the whole 716-instruction stream
    #  is in the file too and runs 1.6 x slower per instruction than any part of it, for a reason that was not found)
    want = (["c2_phase_a_box"] if "c2_phase_a_box" in ratio else ["c2_phase_a_closest"]) if lds_resident else ["c3_node_closest", "c3_node_any"]
    have = [ratio[k][0] for k in want if k in ratio]
    if not have:
        return None
    # the replay was taken on the kernel sources of its round: a note in the file names their hash ("source_hash: ...") -- when the
    # sources have changed since, the factor is still reported, flagged stale (the way pmc_stale flags the counters)
    m = re.search(r"source_hash: (\w+)", open(cands[-1]).read())
    return {"measured_over_priced": round(sum(have) / len(have), 4), "loops": want, "source": os.path.relpath(cands[-1], ROOT),
            "source_hash": m.group(1) if m else None, "stale": (m.group(1) != source_hash()) if m else None}


def cpu_baseline(data, width, height, xs, ys, depth, budget_s=9.0):
    """SURVEY 8d "CPU baseline timing": the CPU oracle (kind "port": a restatement of the reference's algorithm)
    built with the reference's flags (-O3 -ffast-math, CMakeLists.txt:55), scheduled as StaticCamera::render does it
    (persistent threads, 32x32 tile queue, samplesPerPass = 1, one barrier per sample pass), on a BOUNDED sample of the
    same workload: all host cores and one thread, median of 3 runs each."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    import jtx_pathtracer_amd as jtx
    import numpy as np
    capi = jtx._capi
    fast = os.path.join(ROOT, "oracle", "_build", "libjtx_oracle_fast.so")
    src_t = max(os.path.getmtime(os.path.join(ROOT, "oracle", f)) for f in ("jtx_oracle.cpp", "jtx_oracle.h"))
    if not os.path.exists(fast) or os.path.getmtime(fast) < src_t:
        ol.build(fast=True)
    lib = C.CDLL(fast)
    lib.ora_scene_create.restype = C.c_void_p
    lib.ora_scene_create.argtypes = [C.POINTER(capi.SceneDesc)]
    lib.ora_render.restype = None
    lib.ora_render.argtypes = [C.c_void_p, C.POINTER(capi.CameraDesc), C.c_int, C.c_int, C.c_int, C.c_int,
                               C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(capi.Counters)]
    lib.ora_scene_destroy.argtypes = [C.c_void_p]
    desc = data.to_desc()
    h = C.c_void_p(lib.ora_scene_create(C.byref(desc)))
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass

    def run(cam, threads, s0, s1):
        acc = np.zeros((cam.height, cam.width, 3), np.float32)
        img = np.zeros((cam.height, cam.width, 3), np.uint8)
        cnt = capi.Counters()
        t = time.perf_counter()
        lib.ora_render(h, C.byref(cam), threads, s0, s1, 1, acc.ctypes.data_as(C.POINTER(C.c_float)),
                       img.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(cnt))
        return time.perf_counter() - t, cnt.n_closest + cnt.n_any

    def median3(cam, threads, n):
        runs = sorted(run(cam, threads, 1, 1 + n) for _ in range(3))
        dt, rays = runs[1]
        return rays / dt / 1e6, rays, dt

    spp = xs * ys
    cam = data.camera_desc(width, height, xs, ys, depth)
    dt1, _ = run(cam, cores, 0, 1)                        # calibration pass (also warms the caches)
    n = int(max(1, min(spp - 1, budget_s / 3.0 / max(dt1, 1e-3))))
    all_v, all_rays, all_dt = median3(cam, cores, n)
    # one thread: the same view at 1/4 x 1/4 of the pixels (same scene, same rays per pixel), sized the same way
    cam1 = data.camera_desc(max(32, width // 4), max(32, height // 4), xs, ys, depth)
    dt1s, _ = run(cam1, 1, 0, 1)
    n1 = int(max(1, min(spp - 1, budget_s / 3.0 / max(dt1s, 1e-3))))
    one_v, one_rays, one_dt = median3(cam1, 1, n1)
    lib.ora_scene_destroy(h)
    return {"value": round(all_v, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "one_thread": {"value": round(one_v, 3), "unit": "Mrays/s", "cores": 1,
                           "sample": f"median of 3 x strata 1..{n1} of {spp} at {cam1.width}x{cam1.height} ({one_rays} rays, {one_dt:.2f} s)"},
            "sample": f"median of 3 x strata 1..{n} of {spp} of the same {width}x{height} frame ({all_rays} rays, {all_dt:.2f} s per run); "
                      "oracle built -O3 -ffast-math; StaticCamera::render's schedule: persistent threads, 32x32 tile queue, "
                      "samplesPerPass = 1, one barrier per sample pass"}


def roofline_block(workload, scene_info, mine, kernel_name, kernel_ms, launches_per_frame, num_cus, grid_note=""):
    """roofline of the dominant kernel.  The binding ceiling of this path is VALU instruction issue (no MFMA work, and
    the BVH is served from LDS / L2: measured HBM traffic is a few percent of peak), so `bound` names that ceiling and
    `frac` is the issued fraction of it; the HBM view (own-layout algorithmic bytes and PMC-measured bytes) and the
    SURVEY 8d reference-layout figure ride along.  Counter inputs come from profiles/r02_pmc.json (rocprofv3 --pmc
    passes over the same launch, tools/pmc_collect.sh); everything time-like is measured live with HIP events."""
    SIMDS = num_cus * 4
    CLK = 2.4e9                                            # MI355X max shader clock (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE / 8 / kernel time reads 2.39 GHz
    valu_peak = SIMDS * CLK / 2.0 / 1e9                    # wave64 VALU instructions per second: one per 2 cycles per SIMD-32
    lane_peak = num_cus * 128 * CLK / 1e12                 # fp32 lane-operations per second: 4 SIMD-32 per CU (x2 for FMA = the 157 TFLOP/s vector peak)
    t = kernel_ms * 1e-3
    out = {"bound": "valu", "achieved": None, "peak": round(valu_peak, 1), "unit": "G wave-instructions/s", "frac": None,
           "traffic": None, "kernel": kernel_name, "kernel_ms": round(kernel_ms, 4), "launches_per_frame": launches_per_frame,
           "num_cus": num_cus}
    # ---- useful-work fraction: needs no profile, only this frame's device counters and the live kernel time ----
    uops = useful_lane_ops(mine)
    out["useful_frac"] = round(uops / t / 1e12 / lane_peak, 4)
    # ... against what the chip can issue of THAT operation mix with every lane busy: add / sub / mul at 2.4 cycles per wave64 instruction, min /
    # max / compares at 4.4, divisions and square roots at 8.4 -- how far from ATTAINABLE the kernel is, where useful_frac measures against fma rate
    ucyc = useful_issue_cycles(mine)
    out["useful_frac_attainable"] = round(ucyc / (SIMDS * t * CLK), 4)
    out["useful"] = {"lane_ops_per_launch": int(uops), "achieved": round(uops / t / 1e12, 3), "peak": round(lane_peak, 2),
                     "unit": "T fp32 lane-ops/s", "per_event": USEFUL_OPS, "sample_by_class": USEFUL_SAMPLE_CLASS, "eval_by_class": USEFUL_EVAL_CLASS,
                     "issue_cycles_at_least": int(ucyc), "issue_cycles_per_class": list(ISSUE_CYCLES),
                     "note": "reference-algorithm lane-operations (fixed cost per counted event of the untimed counting pass: rays, node visits, "
                             "triangle tests, shading events, camera samples; DESIGN.md section 6) / kernel time / (CUs x 128 lanes x 2.4 GHz)"}
    pmc = None
    path = pmc_file()
    if path:
        try:
            j = json.load(open(path))
            pmc = j.get("workloads", {}).get(workload)
            out["pmc_source"] = os.path.relpath(path, ROOT)
            out["pmc_stale"] = j.get("source_hash") != source_hash()
        except Exception:
            pmc = None
    if pmc and out.get("pmc_stale"):
        # counters of another build say nothing about this one: the instruction-issue figures stay null until
        # tools/pmc_collect.sh has been re-run on the current sources (the useful-work fraction above does not depend on them)
        out["pmc_note"] = "kernel sources changed since the counters were collected: frac / traffic / lane_util withheld"
        pmc = None
    # ---- own-layout algorithmic HBM bytes of this launch ----
    paths = mine["n_camera"]; rays = mine["n_closest"] + mine["n_any"]
    own = 16 * paths                                       # one 16-B radiance record per path (k_resolve_samples reads them back)
    own += (64 + 80) * mine["n_shade"]                     # 64-B shading record + 80-B material per shading event
    ws = (pmc or {}).get("wide_stats")
    if scene_info["lds_resident"]:
        own += scene_info["lds_bytes"] * scene_info.get("workgroups", 0)     # the scene staged once per workgroup
    elif ws:
        own += int(rays * (80 * ws["node_steps_per_ray"] + 32 * ws["leaf_steps_per_ray"] + 48 * ws["tri_tests_per_ray"]))
    hbm = {"algorithmic_bytes_per_launch": int(own), "achieved": round(own / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
           "frac": round(own / t / 1e9 / 8000.0, 4)}
    if pmc:
        c = pmc["counters"]
        insts = c["SQ_INSTS_VALU"]
        out["achieved"] = round(insts / t / 1e9, 1)
        out["frac"] = round(insts / t / 1e9 / valu_peak, 4)
        out["valu_instructions_per_launch"] = int(insts)
        out["lane_util"] = round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 4)
        out["fp32_lane_frac"] = round(out["frac"] * out["lane_util"], 4)       # share of the chip's fp32 lane slots doing work
        # FETCH_SIZE / WRITE_SIZE are KiB; gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md "HBM"): x2 on reads
        measured = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        out["traffic"] = int(measured)
        hbm["measured_bytes_per_launch"] = int(measured)
        hbm["measured_frac"] = round(measured / t / 1e9 / 8000.0, 4)
        hbm["raw_kib"] = {"FETCH_SIZE": c["FETCH_SIZE"], "WRITE_SIZE": c["WRITE_SIZE"]}
        # Issue-cycle model: a wave64 VALU instruction does NOT issue every 2 cycles on gfx950 -- measured with every SIMD holding 8 waves
        # of independent instructions (tools/micro/rate4.hip, profiles/r03_valu_rates.txt): v_add / v_mul / v_fma / v_mov on VGPR or
        # inline-constant operands 2.4 cycles; everything else (min / max / max3, compares, v_cndmask, conversions, bit operations, ANY
        # instruction with an SGPR or literal operand) 4.4; transcendentals 8.4.  Priced like that, the dynamic mix of the launch
        # (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32) gives an UPPER BOUND of the cycles it needs (a forwarded operand makes an instruction
        # cheaper, part of the slow class -- v_and, v_add_u32, a compare + select pair -- costs 2.9-3.8): round 3 read 1.18-1.20 "busy".
        # Round 4 calibrates it on the real thing: the VALU stream of the kernels' own loop bodies (node step of the 8-ary traversal, phase
        # A of the flat leaf list), replayed with its real operands, registers and order by 8 waves per SIMD with nothing else to wait for
        # (tools/isa_replay.py -> profiles/r04_issue_replay.txt), takes 0.83-0.86 of what the classes price it at.  `busy` is the priced
        # demand times that factor over the cycles the SIMDs have: ~1.0 = issue-saturated.
        if c.get("SQ_INSTS_VALU_FMA_F32") is not None:
            fast = c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_FMA_F32"]
            trans = c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
            slow = insts - fast - trans
            need = fast * 2.4 + slow * 4.4 + trans * 8.4
            cal = issue_calibration(bool(scene_info["lds_resident"]))
            out["issue_model"] = {"fast_class_insts": int(fast), "slow_class_insts": int(slow), "transcendental_insts": int(trans),
                                  "cycles_per_inst": {"fast": 2.4, "slow": 4.4, "transcendental": 8.4},
                                  "cycles_priced": int(need), "simd_cycles_available": int(SIMDS * t * CLK),
                                  # `busy`: the class-priced demand over the cycles the SIMDs have -- an upper bound (round 3's meaning of the key,
                                  # kept so that rounds compare; `busy_upper_bound` is the same number under round 4's name);
                                  # `busy_calibrated`: that demand times measured / priced of the kernel's own loop body (round 4 printed this as `busy`)
                                  "busy": round(need / (SIMDS * t * CLK), 3),
                                  "busy_upper_bound": round(need / (SIMDS * t * CLK), 3),
                                  "calibration": cal,
                                  "busy_calibrated": round(need * cal["measured_over_priced"] / (SIMDS * t * CLK), 3) if cal else None,
                                  # round 5 (profiles/r05_box_rates.txt): the two classes are two PIPES that run side by side -- a leaf-box test of 24
                                  # instructions takes 51 cycles where its classes add up to 75 -- so a mix is bound by the larger of (every instruction x
                                  # the 2.13 cycles one issue takes) and (the half-rate instructions x 4.4 + transcendental x 8.4), not by their sum
                                  "co_issue": {"issue_slots_busy": round(insts * 2.13 / (SIMDS * t * CLK), 3),
                                               "half_rate_pipe_busy": round((slow * 4.4 + trans * 8.4) / (SIMDS * t * CLK), 3),
                                               "cycles_per_issue": 2.13, "evidence": "profiles/r05_box_rates.txt"},
                                  "note": "VALU cycles the launch's instruction mix is priced at by class (an upper bound: the classes overlap, see co_issue) x "
                                          "the ratio measured / priced of the kernel's own loop body replayed at saturation (calibration.source), over the cycles "
                                          "the SIMDs have in the kernel's duration (nominal 2.4 GHz)"}
        if c.get("TA_TA_BUSY_sum") and c.get("GRBM_GUI_ACTIVE"):
            # the second ceiling of the HBM-resident kernels: the vector-memory pipeline (one address unit and one data-return unit per CU)
            unit_cycles = num_cus * c["GRBM_GUI_ACTIVE"] / 8.0
            out["vector_memory"] = {"ta_busy": round(c["TA_TA_BUSY_sum"] / unit_cycles, 3), "td_busy": round(c["TD_TD_BUSY_sum"] / unit_cycles, 3),
                                    "load_instructions": int(c["TA_FLAT_READ_WAVEFRONTS_sum"]),
                                    "ta_cycles_per_load": round(c["TA_TA_BUSY_sum"] / max(1.0, c["TA_FLAT_READ_WAVEFRONTS_sum"]), 1)}
        if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            out["l1_hit_rate"] = round(1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
        if c.get("TCC_HIT_sum"):
            out["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            out["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        out["wait_share"] = {"s_waitcnt": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                             "issue_stall": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)}
    out["hbm"] = hbm
    ref_bytes = algorithmic_bytes(mine)
    out["reference_layout_equiv"] = {"bytes_per_launch": int(ref_bytes), "GBps": round(ref_bytes / t / 1e9, 1),
                                     "x_hbm_peak": round(ref_bytes / t / 1e9 / 8000.0, 4),
                                     "note": "SURVEY 8d formula (the reference's 32-B binary nodes / 56-B triangle fetches) on this frame's "
                                             "device counters: what the REFERENCE's layout would have to move; not this kernel's traffic"}
    out["note"] = ("bound = VALU instruction issue: achieved = SQ_INSTS_VALU of this launch (rocprofv3 --pmc, profiles/) / live HIP-event "
                   "kernel time; peak = 4 SIMD x CUs x 2.4 GHz / 2 cycles per wave64 instruction.  lane_util = SQ_THREAD_CYCLES_VALU / "
                   "(64 SQ_ACTIVE_INST_VALU).  useful_frac = reference-algorithm lane-operations / time / fp32 lane peak (issue-independent).  "
                   "hbm.* = own-layout algorithmic bytes and PMC-measured bytes against 8 TB/s." + grid_note)
    return out


def self_launch(n):
    """start `n` ranks of this script under torch.distributed.run on 127.0.0.1 (a free port) and wait for them"""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")                # (what torchrun would set, without its warning on stderr)
    # every rank leaves a note (rank, device, how far it came) in this directory: if the ranks fail, the ONE JSON line this parent prints
    # says how many of them were seen and on which devices, not only "rc != 0" (a driver-run record diagnoses itself)
    import tempfile
    diag = tempfile.mkdtemp(prefix="jtx_bench_ranks_")
    env["JTX_BENCH_DIAG_DIR"] = diag
    proc = subprocess.Popen(cmd, env=env, stderr=subprocess.PIPE, text=True)
    tail = []
    for line in proc.stderr:                              # relayed as it comes; the end of it kept for the record
        sys.stderr.write(line); sys.stderr.flush()
        tail.append(line); del tail[:-120]
    rc = proc.wait()
    if rc != 0:
        print(json.dumps(rank_failure_record(n, rc, diag, "".join(tail))), flush=True)
    import shutil
    shutil.rmtree(diag, ignore_errors=True)
    return rc


def rank_note(stage, **kw):
    """(a rank of a self-launched run) where this rank is: one small file per rank in JTX_BENCH_DIAG_DIR, rewritten at every stage"""
    d = os.environ.get("JTX_BENCH_DIAG_DIR")
    if not d:
        return
    try:
        rank = int(os.environ.get("RANK", "0"))
        with open(os.path.join(d, f"rank{rank}.json"), "w") as f:
            json.dump(dict(rank=rank, local_rank=int(os.environ.get("LOCAL_RANK", "0")), pid=os.getpid(), stage=stage, **kw), f)
    except OSError:
        pass


def rank_wait_for_peers(world, seconds=10.0):
    """(a rank about to stop with an error) give the other ranks time to write their first note: the launcher ends them as soon as one
    rank has failed, and a rank still importing torch would then be missing from the failure record"""
    d = os.environ.get("JTX_BENCH_DIAG_DIR")
    if not d:
        return
    t0 = time.time()
    while time.time() - t0 < seconds and not all(os.path.exists(os.path.join(d, f"rank{r}.json")) for r in range(world)):
        time.sleep(0.05)


def rank_failure_record(n, rc, diag_dir, stderr_tail):
    """the line a failed `bench.py --gpus N` prints instead of a measurement"""
    import glob
    ranks = []
    for f in sorted(glob.glob(os.path.join(diag_dir, "rank*.json"))):
        try:
            ranks.append(json.load(open(f)))
        except (OSError, ValueError):
            pass
    return {"metric": "Mrays/s at 1920x1080x64spp; achieved HBM GB/s vs roofline", "value": None, "unit": "Mrays/s", "n_gpus": n,
            "error": f"the {n} ranks started by bench.py exited with code {rc}", "returncode": rc,
            "nranks_seen": len(ranks), "ranks": sorted(ranks, key=lambda r: r.get("rank", 0)),
            "stderr_tail": stderr_tail[-8000:]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cornell_1920x1080_64spp_d8", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--atrium-tris", type=int, default=262144)
    ap.add_argument("--headline-only", action="store_true", help="skip the other BASELINE.json workloads (N = 1 runs them after the headline)")
    ap.add_argument("--scene", default=None, help="time this asset (.glb / .gltf / .obj, createScene's rules) instead of the procedural scene")
    ap.add_argument("--camera", default=None, help="with --scene: 'inside' or cx,cy,cz,tx,ty,tz,yfov (default: createScene's (0,0,8) -> origin, yfov 20)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL's peer-to-peer needs on this driver
    # The frames in flight live on three HIP streams; the runtime maps streams onto 4 hardware queues by default, and with torch's, the
    # library's and the exchange's streams beside them two render streams then share one -- launches of one queue run in order, and the
    # loop falls back to "one kernel at a time" (C2: 27.2 ms per frame against 26.4; profiles/r05_frames_in_flight.md).  Read by the
    # runtime when it initialises, so it has to be set before torch is imported.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    # JTX_BENCH_FORCE_DIST=1: the N > 1 code path with whatever N is -- also N = 1: the launcher child, an `nccl` process group of one rank,
    # the per-frame exchange through RCCL, the rank diagnosis -- the first contact of that path with real RCCL on a one-GPU box
    force_dist = os.environ.get("JTX_BENCH_FORCE_DIST", "0") not in ("", "0")
    if (args.gpus > 1 or force_dist) and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (the reference needs no launcher either --
        # StaticCamera::render just spawns its workers, camera.cpp:81).  This parent has imported neither torch nor the library and
        # has not touched a GPU: the ranks are a CHILD process (torch.distributed.run), never an exec of a GPU-initialised one; its
        # stdout (rank 0's one JSON line) and its return code are relayed.
        raise SystemExit(self_launch(args.gpus))
    import torch
    import torch.distributed as dist
    import jtx_pathtracer_amd as jtx

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:                    # (WORLD_SIZE=1 set by hand with --gpus N)
            raise SystemExit("--gpus N > 1 with WORLD_SIZE=1: unset WORLD_SIZE (bench.py then starts its own ranks) or launch with torch.distributed.run")
        args.gpus = world
    rank_note("started", devices_visible=torch.cuda.device_count())
    if not torch.cuda.is_available():
        rank_wait_for_peers(world)
        raise SystemExit("bench.py needs an MI355X: no HIP device (there is no CPU fallback)")
    # JTX_DIST_BACKEND=gloo + JTX_ALL_RANKS_ON_DEVICE=0: rehearsal of the N > 1 code path on a one-GPU box (every
    # rank renders its shard on the same card, the exchange is staged through the host); never a measurement
    backend = os.environ.get("JTX_DIST_BACKEND", "nccl")
    if os.environ.get("JTX_ALL_RANKS_ON_DEVICE") is not None:
        local_rank = int(os.environ["JTX_ALL_RANKS_ON_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rank_note("device set", device=local_rank, devices_visible=torch.cuda.device_count())
    dist_on = world > 1 or force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    jtx._capi.check(jtx._capi.load().jtx_mi_set_device(local_rank))
    rank_note("process group up", device=local_rank, backend=backend if dist_on else None)

    wl_name, data, (W, H, xs, ys, depth) = load_workload(jtx, args.workload, args.atrium_tris, args.scene, args.camera)
    t0 = time.perf_counter()
    scene = jtx.Scene(data)
    scene.buildBVH()
    t_upload = time.perf_counter() - t0                     # first creation of the process: includes loading the code object
    t0 = time.perf_counter()
    again = jtx.Scene(data); again.buildBVH()                # what a BVH rebuild after an edit costs (display.cpp:902-905)
    warm_create_ms = round((time.perf_counter() - t0) * 1e3, 2)
    again.destroy()
    cam = data.camera_desc(W, H, xs, ys, depth)
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    img = torch.zeros(H * W * 3, dtype=torch.uint8, device=dev)
    # a non-default torch stream: the kernel launch, the HIP events around it and the RCCL reduce are
    # all ordered on it (stream 0 would mean "the library's own stream" to jtx_mi_render_device)
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    lib = jtx._capi.load()

    integrator = scene.info()["auto_integrator"]
    INTEG_NAMES = {1: "pixel-persistent", 2: "hbm-wavefront"}

    # per-frame exchange: compact own-pixel slabs gathered to rank 0 (default) or one sum-reduce of the full buffers
    collective = os.environ.get("JTX_FRAME_COLLECTIVE", "gather")
    gatherer = jtx.distributed.FrameGather(W, H, rank, world, dev) if (dist_on and collective == "gather") else None

    # N > 1: the exchange of frame i runs on a side stream while frame i + 1 renders into the other pair of shard
    # buffers (rank 0 assembles frames in buffers of their own); every frame is complete before the closing fence
    # ... and, at every N, TWO FRAMES IN FLIGHT: frame i + 1 starts on the other render stream / frame slot while the last chunks and
    # the resolve pass of frame i still run (JTX_FRAMES_IN_FLIGHT=1: one render stream)
    pipe = None
    if (gatherer is not None and os.environ.get("JTX_PIPELINE_EXCHANGE", "1") != "0") or (not dist_on and frames_in_flight() > 1):
        pipe = jtx.distributed.ShardPipeline(scene, cam, rank, world, dev, gatherer, integrator=integrator, timing=dist_on,
                                             frames_in_flight=frames_in_flight(), deliver_to_host=True)

    xtimed = []                                              # serial exchange: (event before, event after) per frame, on the render stream

    def step(count=False, profile=False, last=False):
        if pipe is not None and not count and not profile:
            pipe.step(last=last)
            return
        jtx.distributed.render_shard(scene, cam, rank, world, acc, img, stream=stream, count_rays=count,
                                     integrator=integrator, profile_kernels=profile)
        if dist_on:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(tstream)
        if gatherer is not None:
            gatherer.collect(acc, img)
        else:
            jtx.distributed.reduce_frame(acc, img, dst=0)
        if dist_on:
            e1.record(tstream)
            xtimed.append((e0, e1))

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # The exchange form is decided BEFORE the first frame: JTX_FRAME_COLLECTIVE names it, and a one-element probe of the
    # slab gather (every rank, same order) falls back to the reduce on all ranks together if this RCCL build refuses it.
    if gatherer is not None and backend == "nccl":
        ok = 1
        try:
            probe = torch.zeros(4, dtype=torch.uint8, device=dev)
            dist.gather(probe, [torch.zeros_like(probe) for _ in range(world)] if rank == 0 else None, dst=0)
            torch.cuda.synchronize()
        except RuntimeError as e:
            sys.stderr.write(f"[bench rank {rank}] gather probe failed ({e}); falling back to reduce\n")
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            gatherer, pipe, collective = None, None, "reduce"
    # counted pass: deterministic ray / node / triangle tallies of this rank's shard (untimed)
    step(count=True)
    fence()
    cnt = jtx._capi.Counters()
    jtx._capi.check(lib.jtx_mi_get_counters(scene.handle, C.byref(cnt)))
    mine = cnt.as_dict()
    flat = flat_counters(mine)
    keys = sorted(flat)
    tot = torch.tensor([flat[k] for k in keys], dtype=torch.int64, device=dev)
    if dist_on:
        dist.all_reduce(tot)
    total = unflat_counters(dict(zip(keys, [int(v) for v in tot.tolist()])))
    rays_frame = total["n_closest"] + total["n_any"]

    # wavefront: one extra untimed frame with a HIP event pair around every kernel, to find the dominant stage
    kind_ms = None
    if integrator == 2:
        step(profile=True)
        fence()
        kms = (C.c_float * 5)(); kn = (C.c_int32 * 5)()
        jtx._capi.check(lib.jtx_mi_kernel_time_by_kind(scene.handle, kms, kn))
        kind_ms = [(float(kms[i]), int(kn[i])) for i in range(5)]

    if pipe is not None:
        pipe.prime()                                         # set-up: every frame slot's buffers exist and are mapped (untimed, like scene creation)
    for _ in range(args.warmup):
        step()
    fence()
    ms = C.c_float(); nl = C.c_int32()
    jtx._capi.check(lib.jtx_mi_kernel_time(scene.handle, C.byref(ms), C.byref(nl)))   # drop warm-up events
    del xtimed[:]
    if pipe is not None:
        pipe.reset_timing()
    t = time.perf_counter()
    for i in range(args.steps):
        step(last=(i == args.steps - 1))
    fence()
    elapsed = time.perf_counter() - t
    rank_note("timed region done", device=local_rank, steps=args.steps)
    delivered = pipe is not None and getattr(pipe, "deliver", False)
    if delivered:                                            # the copies arrived: the last frame, word for word
        nbuf = len(pipe.accs); lastb = (pipe.n - 1) % nbuf
        src_acc, src_img = (pipe.accs[lastb], pipe.imgs[lastb]) if pipe.gatherer is None else (pipe.frame_acc, pipe.frame_img)
        if not (torch.equal(pipe.host_acc[lastb], src_acc.cpu()) and torch.equal(pipe.host_img[lastb], src_img.cpu())):
            raise SystemExit("bench: the host copy of the last frame differs from the device film")
    jtx._capi.check(lib.jtx_mi_kernel_time(scene.handle, C.byref(ms), C.byref(nl)))
    kernel_ms_timed = ms.value / max(1, nl.value)          # launches of the timed region: start-to-end, overlapping when frames are in flight
    in_flight = len(pipe.rstreams) if pipe is not None else 1
    # the roofline's kernel duration: launches with ONE frame in flight (integrator 1; the wavefront's dominant stage is timed per kind above)
    kernel_ms = serial_kernel_ms(jtx, torch, lib, scene, cam, rank, world, dev, integrator, frames=min(3, args.steps)) if (in_flight > 1 and integrator == 1) else kernel_ms_timed

    tt = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms_max = float(tt[0]), float(tt[1])
    # self-diagnosis of the first real multi-GPU run: how many ranks took part (an all-reduce of ones), which device each rank
    # rendered on and what its shard kernel took
    ranks_diag = None
    if dist_on:
        ddev = dev if backend == "nccl" else torch.device("cpu")          # (gloo rehearsals: host tensors)
        ones = torch.ones(1, dtype=torch.int32, device=ddev)
        dist.all_reduce(ones)
        # this rank's exchange (pack + gather / reduce + scatter): device time between events on the stream it runs on
        if pipe is not None and gatherer is not None:
            exchange_ms = pipe.exchange_ms()
        else:
            exchange_ms = sum(a.elapsed_time(b) for a, b in xtimed) / len(xtimed) if xtimed else None
        rowt = torch.tensor([float(rank), float(torch.cuda.current_device()), kernel_ms, float(mine["n_closest"] + mine["n_any"]),
                             -1.0 if exchange_ms is None else float(exchange_ms)], dtype=torch.float64, device=ddev)
        rows = [torch.zeros_like(rowt) for _ in range(world)]
        dist.all_gather(rows, rowt)
        ranks_diag = {"nranks_seen": int(ones.item()),
                      "ranks": [{"rank": int(r[0]), "device": int(r[1]), "shard_kernel_ms": round(float(r[2]), 4), "shard_rays": int(r[3]),
                                 "exchange_ms": None if r[4] < 0 else round(float(r[4]), 4)}
                                for r in (x.tolist() for x in rows)]}

    # N = 1, beside the headline: (a) the same loop WITHOUT the host copies (round 5's timed region: film buffers resident in HBM); (b) the same
    # frame through the blocking jtx_mi_render (the reference's API shape: Camera::render returns with the film on the host; one frame at
    # a time); (c) the frame as the reference's UI renders it -- a callback after every pass of samplesPerPass_ strata (camera.hpp:181:
    # the default is 1) -- through jtx_mi_render's progressive launch
    device_ms = host_ms = None
    progressive = {}
    if not dist_on:
        import numpy as np
        if integrator == 1 and pipe is not None:
            dpipe = jtx.distributed.ShardPipeline(scene, cam, rank, world, dev, None, integrator=integrator, frames_in_flight=frames_in_flight())
            nd = max(1, min(args.steps, 10))
            for _ in range(len(dpipe.accs)):
                dpipe.step()
            torch.cuda.synchronize()
            td = time.perf_counter()
            for i in range(nd):
                dpipe.step(last=(i == nd - 1))
            torch.cuda.synchronize()
            device_ms = (time.perf_counter() - td) / nd * 1e3
            del dpipe
        hacc = np.zeros(H * W * 3, np.float32); himg = np.zeros(H * W * 3, np.uint8)
        for a in (hacc, himg):                        # page-locked like Camera::acc_ / img_ in the host mirrors (jtx_mi_pin_host)
            lib.jtx_mi_pin_host(a.ctypes.data_as(C.c_void_p), a.nbytes)
        o = jtx._capi.RenderOpts(); o.integrator = integrator

        def host_frame(cb=jtx._capi.PROGRESS_CB(0)):
            jtx._capi.check(lib.jtx_mi_render(scene.handle, C.byref(cam), C.byref(o), hacc.ctypes.data_as(C.POINTER(C.c_float)),
                                              himg.ctypes.data_as(C.POINTER(C.c_uint8)), cb, None))
        host_frame()
        th = time.perf_counter()
        nh = max(1, min(args.steps, 5))
        for _ in range(nh):
            host_frame()
        host_ms = (time.perf_counter() - th) / nh * 1e3
        if integrator == 1 and not args.headline_only:
            ncb = [0]
            cbf = jtx._capi.PROGRESS_CB(lambda cur, tot, user: (ncb.__setitem__(0, ncb[0] + 1), 0)[1])
            for spp_pass in (1, 8):
                o.samples_per_tick = spp_pass
                host_frame(cbf)
                ncb[0] = 0
                th = time.perf_counter()
                for _ in range(nh):
                    host_frame(cbf)
                pms = (time.perf_counter() - th) / nh * 1e3
                progressive[f"{wl_name}@spp_per_pass_{spp_pass}"] = {
                    "ms_per_step": round(pms, 3), "value": round(rays_frame / pms / 1e3, 2), "unit": "Mrays/s", "steps": nh,
                    "samples_per_pass": spp_pass, "callbacks_per_frame": ncb[0] // nh, "rays_per_frame": rays_frame,
                    "timed_region": "jtx_mi_render with a callback per pass: one launch for all passes, a resolver kernel beside it; the preview "
                                    "and, at the end, the film delivered to page-locked host buffers"}
            o.samples_per_tick = 0
        for a in (hacc, himg):
            lib.jtx_mi_unpin_host(a.ctypes.data_as(C.c_void_p))
        jtx._capi.check(lib.jtx_mi_kernel_time(scene.handle, C.byref(ms), C.byref(nl)))     # drop those events

    if rank == 0:
        value = rays_frame * args.steps / elapsed / 1e6
        kernel_name = "k_render_paths"
        launches_per_frame = 1
        info = scene.info()
        sinfo = {"lds_resident": bool(info["lds_resident"]), "lds_bytes": 8 * 32 * info["num_nodes"] + 48 * info["num_prims"],
                 "workgroups": info["resident_workgroups"]}
        roof_counters = mine
        if integrator == 2:
            names = ["k_wf_generate", "k_wf_trace<closest>", "k_wf_shade", "k_wf_trace<any>", "k_wf_resolve"]
            dom = max(range(5), key=lambda i: kind_ms[i][0])
            kernel_name = names[dom]
            launches_per_frame = max(1, kind_ms[dom][1])
            kernel_ms = kind_ms[dom][0] / launches_per_frame          # average launch duration of that kernel
        key = wl_name if world == 1 else wl_name + f"@{world}"
        roof = roofline_block(key, sinfo, roof_counters, kernel_name, kernel_ms, launches_per_frame, info["num_cus"])
        if in_flight > 1 and integrator == 1 and not dist_on:
            # ONE self-consistent pair: `value` and the roofline on the SAME time -- the wall time per frame of the timed
            # region (frames in flight: a frame's share of the chip; kernel_ms <= ms_per_step by construction).  The figures of a LONE launch
            # (HIP events, one frame in flight: what the committed rocprofv3 kernel statistics show) ride under `lone`.
            lone = roof
            roof = roofline_block(key, sinfo, roof_counters, kernel_name, elapsed / args.steps * 1e3, launches_per_frame, info["num_cus"])
            roof["time_basis"] = ("wall time per frame of the timed region: %d frames in flight, every film delivered to the host -- the chip runs nothing "
                                  "but these frames, so a frame's share of it is the time between two frames" % in_flight)
            roof["frames_in_flight"] = in_flight
            roof["launch_ms_start_to_end"] = round(kernel_ms_timed, 3)
            roof["lone"] = {k: lone[k] for k in ("kernel_ms", "frac", "achieved", "useful_frac", "useful_frac_attainable", "issue_model", "wait_share",
                                                 "vector_memory", "lane_util", "fp32_lane_frac") if k in lone}
            roof["lone"]["hbm_measured_frac"] = lone["hbm"].get("measured_frac")
            roof["lone"]["note"] = ("the same numerators over the average duration of a launch with ONE frame in flight (HIP events on the launch stream; "
                                    "`JTX_FRAMES_IN_FLIGHT=1 bench.py` times these, profiles/ holds their rocprofv3 statistics)")
        elif in_flight > 1 and integrator == 1:
            roof["in_flight"] = in_flight_block(roof, in_flight, elapsed / args.steps * 1e3, kernel_ms_timed)
        out = {
            "metric": "Mrays/s at 1920x1080x64spp; achieved HBM GB/s vs roofline",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl_name, "scene_triangles": data.num_triangles, "width": W, "height": H,
                       "spp": xs * ys, "max_depth": depth, "rays_per_frame": rays_frame,
                       "rays_per_sample": round(rays_frame / max(1, total["n_camera"]), 4),
                       "parallelism": (f"pixel-tile shard x{world} + 1 {collective}/frame" + (" overlapped with the next frame" if pipe is not None and gatherer is not None else "") + ("" if backend == "nccl" else f" (REHEARSAL over {backend})") if dist_on else "1 gpu")
                                      + (f", {len(pipe.rstreams)} frames in flight" if pipe is not None and len(pipe.rstreams) > 1 else ""),
                       "scene_upload_ms": round(t_upload * 1e3, 2), "scene_create_warm_ms": warm_create_ms,
                       "integrator": INTEG_NAMES[integrator], "lds_resident_bvh": info["lds_resident"],
                       "wide_bvh_bytes": info["wide_bytes"],
                       "frames_in_flight": in_flight,
                       "timed_region": ("frames in flight; every film (accumulation buffer + RGB8 image) delivered to page-locked host memory"
                                        if delivered else "frames rendered into HBM-resident film buffers (jtx_mi_render_device), incl. the resolve pass")},
            "roofline": roof,
        }
        if ranks_diag is not None:
            out["ranks"] = ranks_diag
        if device_ms is not None:
            out["ms_per_step_device"] = round(device_ms, 3)                 # the same loop, films left in HBM (round 5's timed region)
            out["value_device"] = round(rays_frame / device_ms / 1e3, 2)
        if host_ms is not None:
            out["ms_per_step_host_blocking"] = round(host_ms, 3)            # jtx_mi_render: one frame at a time, returns with the film on the host
            out["value_host_blocking"] = round(rays_frame / host_ms / 1e3, 2)
        if not dist_on and not args.headline_only and args.scene is None:
            # the other BASELINE.json workloads, a few frames each (C3 / C5 / C1 belong in the driver-written record);
            # outside `value`, `steps`, `ms_per_step`, which stay the headline's
            extras = {}
            for name in DEFAULT_EXTRAS + ("cornell_1920x1080_64spp_d8",):
                if name == args.workload:
                    continue
                try:
                    n2, d2, dims2 = load_workload(jtx, name, args.atrium_tris)
                    extras[n2] = time_workload(jtx, torch, dev, tstream, n2, d2, dims2, EXTRA_STEPS.get(name, 3), 1)
                except Exception as e:                    # report, never hide
                    extras[name] = {"failed": f"{type(e).__name__}: {e}"}
            # the architecture north_star names -- integrator 2, the HBM wavefront (SoA queues, ballot compaction; with JTX_WF_SORT_SHADE=1
            # one shade launch per Material::type: the material-sorted queues) -- on the workload it was named for, as a CURRENT figure
            # beside the integrator that ships.  Same film bit for bit (tests); launches of one scene are serialised.
            for tag, sort in (("wavefront", "0"), ("wavefront_sorted", "1")):
                prev = os.environ.get("JTX_WF_SORT_SHADE")
                os.environ["JTX_WF_SORT_SHADE"] = sort
                try:
                    n2, d2, dims2 = load_workload(jtx, "mixed_1920x1080_128spp_d8", args.atrium_tris)
                    e = time_workload(jtx, torch, dev, tstream, n2, d2, dims2, 3, 1, integrator=2)
                    e["integrator_name"] = "hbm-wavefront" + (", one shade launch per material type" if sort == "1" else "")
                    extras[f"{n2}@{tag}"] = e
                except Exception as e:                    # report, never hide
                    extras[f"mixed_1920x1080_128spp_d8@{tag}"] = {"failed": f"{type(e).__name__}: {e}"}
                finally:
                    if prev is None:
                        os.environ.pop("JTX_WF_SORT_SHADE", None)
                    else:
                        os.environ["JTX_WF_SORT_SHADE"] = prev
            for wfname, steps_wf in (("cornell_1920x1080_64spp_d8", 5), ("atrium_1920x1080_64spp_d8", 2)):      # (current figures)
                try:
                    n2, d2, dims2 = load_workload(jtx, wfname, args.atrium_tris)
                    e = time_workload(jtx, torch, dev, tstream, n2, d2, dims2, steps_wf, 1, integrator=2)
                    e["integrator_name"] = "hbm-wavefront"
                    extras[f"{n2}@wavefront"] = e
                except Exception as e:                    # report, never hide
                    extras[f"{wfname}@wavefront"] = {"failed": f"{type(e).__name__}: {e}"}
            extras.update(progressive)
            out["workloads"] = extras
        if not dist_on and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(data, W, H, xs, ys, depth)
            except Exception as e:                        # report, never hide
                out["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
