#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_golden.json from the CPU oracle.

These vectors are produced by THIS repo's oracle (oracle/jtx_oracle.cpp), not by the reference: the
reference has no tests or fixtures and cannot be built here (un-vendored jtxlib).
They pin the oracle against accidental drift.  The vectors that DO come from the reference's own
sources are the check values recorded in SURVEY.md (probe run of the unmodified reference sources):
see REFERENCE_PROBE below; test_oracle_cpu.py checks the oracle against them.
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol           # noqa: E402
import jtx_pathtracer_amd as jtx  # noqa: E402

# Values measured on the reference's own hot-path sources by the survey
REFERENCE_PROBE = {
    "quad_64x64_2x2_d4_rgb8_fnv1a": "1af9ba89",
    "cornell_512_4x4_d4_closest_Mrays": 15.27, "cornell_512_4x4_d4_any_Mrays": 11.07,
    "cornell_512_4x4_d4_rays_per_sample": 6.28,
    "cornell_1920x1080_2x2_d8_closest_Mrays": 26.9, "cornell_1920x1080_2x2_d8_any_Mrays": 18.6,
    "cornell_1920x1080_2x2_d8_rays_per_sample": 5.49,
    "cornell_bvh_nodes": 55, "cornell_bvh_leaves": 28, "cornell_bvh_depth": 6,
}


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def kernel_inputs():
    """seeded inputs of the per-function known answers (SURVEY.md 8c ii, iii, vii); shared with the tests"""
    rs = np.random.RandomState(77)
    n = 4096
    # (ii) AABB::hit: random boxes / rays, a quarter axis-parallel (zero direction components), some starting on a face
    lo = rs.uniform(-5, 5, (n, 3)).astype(np.float32)
    hi = (lo + rs.uniform(0, 4, (n, 3)).astype(np.float32)).astype(np.float32)
    o = rs.uniform(-8, 8, (n, 3)).astype(np.float32)
    d = rs.normal(size=(n, 3)).astype(np.float32)
    aim = (lo + (hi - lo) * rs.uniform(-0.2, 1.2, (n, 3)).astype(np.float32) - o).astype(np.float32)
    d[n // 2:] = aim[n // 2:]                                            # half the rays are aimed at (or just past) their box
    z = rs.randint(0, 3, n)
    d[np.arange(n)[: n // 4], z[: n // 4]] = 0.0
    d[np.arange(n)[: n // 8], (z[: n // 8] + 1) % 3] = -0.0
    o[n // 16: n // 8, 0] = lo[n // 16: n // 8, 0]                       # origin exactly on the min-x plane
    t1 = rs.uniform(0.5, 30, n).astype(np.float32)
    # (iii) ray-triangle: Cornell rays aimed at triangle vertices, edge midpoints and interiors (grazers included)
    return dict(lo=lo, hi=hi, o=o, d=d, t1=t1)


def tri_rays(data):
    rs = np.random.RandomState(78)
    V = []
    for m in data.meshes:
        v = m["vertices"][m["indices"]]                                   # (F, 3, 3)
        V.append(v)
    v = np.concatenate(V)
    pts = np.concatenate([v.reshape(-1, 3), (v[:, 0] + v[:, 1]) / 2, (v[:, 1] + v[:, 2]) / 2, v.mean(1)]).astype(np.float32)
    pts = np.repeat(pts, 8, axis=0)
    o = (np.array([[278.0, 273.0, -800.0]], np.float32) + rs.uniform(-250, 250, (len(pts), 3)).astype(np.float32) *
         np.array([[1, 1, 0]], np.float32)).astype(np.float32)
    o[::2] = np.array([278.0, 273.0, 279.5], np.float32) + rs.uniform(-200, 200, (len(o[::2]), 3)).astype(np.float32)
    return o, (pts - o).astype(np.float32)


def kernel_vectors(scenes):
    k = kernel_inputs()
    lib = ol.load()
    import ctypes as C
    hits = np.zeros(len(k["lo"]), np.uint8)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    lib.ora_aabb_hit.restype = C.c_int
    for i in range(len(hits)):
        lo, hi, o, d = (np.ascontiguousarray(k[x][i]) for x in ("lo", "hi", "o", "d"))
        hits[i] = lib.ora_aabb_hit(fp(lo), fp(hi), fp(o), fp(d), C.c_float(0.001), C.c_float(float(k["t1"][i])))
    out = {"aabb_hit_crc32": crc(hits), "aabb_hit_count": int(hits.sum())}
    cor = ol.OracleScene(scenes["cornell"])
    o, d = tri_rays(scenes["cornell"])
    r = cor.closestHit(o, d)
    out["cornell_closest"] = {"rays": len(o), "hits": int(r["hit"].sum()),
                              "crc32": crc(np.concatenate([r["hit"].astype(np.float32), r["t"], r["prim"].astype(np.float32), r["b1"], r["b2"],
                                                           r["point"].reshape(-1), r["normal"].reshape(-1), r["uv"].reshape(-1)]))}
    a = cor.anyHit(o, d, 0.0, np.full(len(o), 0.9999, np.float32))
    out["cornell_any"] = {"hits": int(a.sum()), "crc32": crc(a)}
    nodes, refs = cor.bvh()
    out["cornell_bvh_crc32"] = crc(nodes.tobytes() + refs.tobytes())
    # (vii) per-sample radiance: 256 fixed pixels x 16 samples, Cornell 512x512, 4x4 strata, depth 4 (config C1)
    rs = np.random.RandomState(79)
    row = np.repeat(rs.randint(0, 512, 256), 16).astype(np.int32); col = np.repeat(rs.randint(0, 512, 256), 16).astype(np.int32)
    smp = np.tile(np.arange(16), 256).astype(np.int32)
    rgb = cor.radiance_samples(scenes["cornell"].camera_desc(512, 512, 4, 4, 4), row, col, smp)
    out["cornell_radiance_samples_crc32"] = crc(rgb)
    return out


def main():
    out = {"reference_probe": REFERENCE_PROBE, "rng": {}, "sincos": {}, "renders": {}, "bxdf": {}}
    for seed in [(0, 0, 1), (1, 2, 3), (511, 17, 16), (1079, 1919, 64)]:
        u, f = ol.rng_stream(*seed, 16)
        out["rng"]["%d,%d,%d" % seed] = {"u32": [int(x) for x in u], "f32_bits": [int(x) for x in f.view(np.uint32)]}
    x = np.array([0.0, 0.5, 1.0, np.pi / 4, np.pi / 2, 2.0, 3.0, np.pi, 4.0, 5.5, 2 * np.pi, -0.7], np.float32)
    s, c = ol.sincos(x)
    out["sincos"] = {"x_bits": [int(v) for v in x.view(np.uint32)], "sin_bits": [int(v) for v in s.view(np.uint32)],
                     "cos_bits": [int(v) for v in c.view(np.uint32)]}
    scenes = {"cornell": jtx.scenes.cornell(), "quad": jtx.scenes.quad_scene(), "mixed": jtx.scenes.mixed(sphere_res=(16, 8))}
    for name, (w, h, xs, ys, d) in {"cornell": (96, 64, 2, 2, 4), "quad": (64, 64, 2, 2, 4), "mixed": (80, 60, 2, 1, 6)}.items():
        o = ol.OracleScene(scenes[name])
        acc, img, cnt = o.render(scenes[name].camera_desc(w, h, xs, ys, d))
        out["renders"][name] = {"size": [w, h, xs, ys, d], "acc_crc32": crc(acc), "img_crc32": crc(img), "counters": cnt,
                                "bvh": o.info()}
    # BxDF known answers: 8 inputs per material of the mixed scene
    rs = np.random.RandomState(5)
    n = 8
    nrm = rs.normal(size=(n, 3)).astype(np.float32); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
    wo = rs.normal(size=(n, 3)).astype(np.float32); wo /= np.linalg.norm(wo, axis=1, keepdims=True).astype(np.float32)
    wi = rs.normal(size=(n, 3)).astype(np.float32); wi /= np.linalg.norm(wi, axis=1, keepdims=True).astype(np.float32)
    uc = rs.uniform(0, 1, n).astype(np.float32); u2 = rs.uniform(0, 1, (n, 2)).astype(np.float32); uv = rs.uniform(0, 1, (n, 2)).astype(np.float32)
    o = ol.OracleScene(scenes["mixed"])
    out["bxdf"]["inputs"] = {k: [int(v) for v in a.reshape(-1).view(np.uint32)] for k, a in
                             dict(normal=nrm, wo=wo, wi=wi, uc=uc, u2=u2, uv=uv).items()}
    out["bxdf"]["materials"] = {}
    for m in range(len(scenes["mixed"].materials)):
        sm = o.sampleBxdf(m, nrm, wo, uc, u2, uv)
        out["bxdf"]["materials"][str(m)] = {
            "sample_crc32": crc(np.concatenate([sm["ok"].astype(np.float32), sm["f"].reshape(-1), sm["wi"].reshape(-1), sm["pdf"]])),
            "eval_crc32": crc(o.evalBxdf(m, nrm, wo, wi, uv)), "pdf_crc32": crc(o.pdfBxdf(m, nrm, wo, wi, uv))}
    out["kernels"] = kernel_vectors(scenes)
    with open(os.path.join(HERE, "oracle_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote oracle_golden.json")


if __name__ == "__main__":
    main()
