"""CPU tests (no GPU): the oracle against known answers, the reference-run check values recorded in
SURVEY.md, the committed golden vectors; the host BVH builder against the oracle's; the C-ABI's
exported symbols.  Whole file runs in well under a minute on 8 cores.
"""
import ctypes as C
import json
import os
import re
import zlib

import numpy as np
import pytest

import oracle_lib as ol
import jtx_pathtracer_amd as jtx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden.json")))
PROBE = GOLD["reference_probe"]


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------
# RNG: an independent pure-Python statement of util/rand.hpp
# ------------------------------------------------------------------------------------------------
M32 = 0xFFFFFFFF


def py_fnv1a_3(x, y, n):
    h = 2166136261
    for v in (x, y, n):
        h ^= v & M32
        h = (h * 16777619) & M32
    return h


class PyRng:
    def __init__(self, x, y, n):
        self.state = 0                      # quirk Q1: state_ read as 0
        self.advance()
        self.state = (self.state + py_fnv1a_3(x, y, n)) & M32
        self.advance()

    def advance(self):
        s = self.state
        self.state = (s * 747796405 + 2891336453) & M32
        word = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & M32
        return ((word >> 2) ^ word) & M32   # NB ">> 2", not PCG's ">> 22" (rand.hpp:103)


@pytest.mark.parametrize("seed", [(0, 0, 1), (1, 2, 3), (511, 17, 16), (1079, 1919, 64), (4000000000, 7, 9)])
def test_rng_matches_independent_python(seed):
    u, f = ol.rng_stream(*seed, 32)
    r = PyRng(*seed)
    ref = [r.advance() for _ in range(32)]
    assert [int(x) for x in u] == ref
    assert np.array_equal(f, np.array([(x & 0xFFFFFF) / 16777216.0 for x in ref], np.float32))
    assert ol.load().ora_fnv1a_3(*seed) == py_fnv1a_3(*seed)


def test_rng_golden_vectors():
    for key, g in GOLD["rng"].items():
        seed = tuple(int(v) for v in key.split(","))
        u, f = ol.rng_stream(*seed, 16)
        assert [int(x) for x in u] == g["u32"]
        assert [int(x) for x in f.view(np.uint32)] == g["f32_bits"]


def test_sample_range_quirk():
    """sampleRange(n-1): exactly one advance, hi32(x*(n-1)); n == 1 -> index 0 (quirk Q2)."""
    lib = ol.load()
    for n_lights in (1, 2, 3, 5):
        r = PyRng(3, 4, 5)
        x = r.advance()
        expect = 0 if n_lights - 1 <= 0 else (x * (n_lights - 1)) >> 32
        assert lib.ora_rng_sample_range(3, 4, 5, 0, n_lights - 1) == expect
        assert expect <= max(0, n_lights - 2)           # the last light is never chosen


# ------------------------------------------------------------------------------------------------
# deterministic sin/cos
# ------------------------------------------------------------------------------------------------
def test_sincos_accuracy_and_golden():
    x = np.linspace(-1.0, 7.0, 200001).astype(np.float32)
    s, c = ol.sincos(x)
    # within 2 ulp of the correctly rounded value over the range the warps use ([-pi/4, 2*pi])
    for got, ref in ((s, np.sin(x.astype(np.float64))), (c, np.cos(x.astype(np.float64)))):
        err = np.abs(got.astype(np.float64) - ref)
        ulp = np.spacing(np.maximum(np.abs(ref), 2.0 ** -20).astype(np.float32)).astype(np.float64)
        assert (err / ulp).max() < 2.5
    g = GOLD["sincos"]
    xg = np.array(g["x_bits"], np.uint32).view(np.float32)
    sg, cg = ol.sincos(xg)
    assert [int(v) for v in sg.view(np.uint32)] == g["sin_bits"]
    assert [int(v) for v in cg.view(np.uint32)] == g["cos_bits"]


def test_det_epsilon_equivalence():
    """mesh.hpp:114 compares |det| < 1e-8 in double; the kernels use |det| <= 1e-8f."""
    e = np.float32(1e-8)
    assert float(e) < 1e-8 < float(np.nextafter(e, np.float32(1)))


# ------------------------------------------------------------------------------------------------
# AABB::hit against an independent numpy statement of aabb.hpp:66-81
# ------------------------------------------------------------------------------------------------
def np_aabb_hit(pmin, pmax, o, d, t0, t1):
    f = np.float32
    t0, t1 = f(t0), f(t1)
    with np.errstate(all="ignore"):
        for i in range(3):
            inv = f(1.0) / f(d[i])
            tn = f((f(pmin[i]) - f(o[i])) * inv)
            tf = f((f(pmax[i]) - f(o[i])) * inv)
            if tn > tf:
                tn, tf = tf, tn
            t0 = tn if tn > t0 else t0
            t1 = tf if tf < t1 else t1
            if t0 > t1:
                return False
    return True


def test_aabb_hit_random_and_axis_parallel():
    lib = ol.load()
    rs = np.random.RandomState(2)
    f3 = C.c_float * 3
    n_hit = 0
    for k in range(3000):
        a, b = rs.uniform(-5, 5, 3).astype(np.float32), rs.uniform(-5, 5, 3).astype(np.float32)
        pmin, pmax = np.minimum(a, b), np.maximum(a, b)
        o = rs.uniform(-8, 8, 3).astype(np.float32)
        d = rs.normal(size=3).astype(np.float32)
        if k % 2 == 0:
            d = ((pmin + pmax) / 2 - o + rs.normal(size=3)).astype(np.float32)     # aimed at the box
        if k % 5 == 0:
            d[rs.randint(3)] = 0.0                      # axis-parallel: 1/d = inf, 0*inf = NaN paths
        if k % 7 == 0:
            o[rs.randint(3)] = pmin[rs.randint(3)]      # origin on a slab plane
        if k % 11 == 0:
            d[rs.randint(3)] = -0.0
        got = lib.ora_aabb_hit(f3(*pmin), f3(*pmax), f3(*o), f3(*d), C.c_float(0.001), C.c_float(np.inf))
        assert bool(got) == np_aabb_hit(pmin, pmax, o, d, 0.001, np.inf), (pmin, pmax, o, d)
        n_hit += got
    assert 50 < n_hit < 2700


# ------------------------------------------------------------------------------------------------
# The oracle against what the survey measured on the reference's own sources
# ------------------------------------------------------------------------------------------------
def fnv1a_bytes(b):
    h = 2166136261
    for x in b:
        h ^= x
        h = (h * 16777619) & M32
    return h


def test_reference_probe_quad_image_hash():
    """createMeshScene quad, 64x64, 2x2 spp, depth 4 -> FNV-1a(RGB8) = 1af9ba89
    on the reference's unmodified hot-path sources (state_=0 reading).  The oracle must reproduce it,
    with the deterministic sin/cos and with libm's."""
    q = jtx.scenes.quad_scene()
    o = ol.OracleScene(q)
    cam = q.camera_desc(64, 64, 2, 2, 4)
    for mode in (0, 1):
        ol.load().ora_set_sincos_mode(mode)
        try:
            _, img, _ = o.render(cam)
        finally:
            ol.load().ora_set_sincos_mode(0)
        assert "%08x" % fnv1a_bytes(img.tobytes()) == PROBE["quad_64x64_2x2_d4_rgb8_fnv1a"]


def test_reference_probe_cornell_bvh():
    o = ol.OracleScene(jtx.scenes.cornell())
    nodes, refs = o.bvh()
    assert len(nodes) == PROBE["cornell_bvh_nodes"]
    assert int((nodes["num_prims"] > 0).sum()) == PROBE["cornell_bvh_leaves"]
    assert o.info()["max_depth"] == PROBE["cornell_bvh_depth"]
    assert nodes["num_prims"].max() <= 2 and len(refs) == 32


def test_reference_probe_cornell_ray_counts_config1():
    """BASELINE config 1 (Cornell 512x512, 4x4 spp, depth 4): the reference traced 15.27 M closestHit +
    11.07 M anyHit calls, 6.28 rays/sample."""
    data = jtx.scenes.cornell()
    _, _, cnt = ol.OracleScene(data).render(data.camera_desc(512, 512, 4, 4, 4))
    # the survey prints two decimals (rounded or truncated): agree to the printed precision
    assert abs(cnt["n_closest"] / 1e6 - PROBE["cornell_512_4x4_d4_closest_Mrays"]) < 0.01
    assert abs(cnt["n_any"] / 1e6 - PROBE["cornell_512_4x4_d4_any_Mrays"]) < 0.01
    assert abs((cnt["n_closest"] + cnt["n_any"]) / cnt["n_camera"] - PROBE["cornell_512_4x4_d4_rays_per_sample"]) < 0.01
    assert cnt["n_shade"] == cnt["n_any"]        # one shadow ray per shading event (one light, always sampled)


def test_reference_probe_cornell_ray_counts_1080p():
    data = jtx.scenes.cornell()
    _, _, cnt = ol.OracleScene(data).render(data.camera_desc(1920, 1080, 2, 2, 8))
    assert abs(cnt["n_closest"] / 1e6 - PROBE["cornell_1920x1080_2x2_d8_closest_Mrays"]) < 0.1
    assert abs(cnt["n_any"] / 1e6 - PROBE["cornell_1920x1080_2x2_d8_any_Mrays"]) < 0.1
    assert abs((cnt["n_closest"] + cnt["n_any"]) / cnt["n_camera"] - PROBE["cornell_1920x1080_2x2_d8_rays_per_sample"]) < 0.01


# ------------------------------------------------------------------------------------------------
# committed golden vectors (drift detection) and oracle self-consistency
# ------------------------------------------------------------------------------------------------
def _scene(name):
    return {"cornell": jtx.scenes.cornell, "quad": jtx.scenes.quad_scene,
            "mixed": lambda: jtx.scenes.mixed(sphere_res=(16, 8))}[name]()


@pytest.mark.parametrize("name", sorted(GOLD["renders"]))
def test_render_golden(name):
    g = GOLD["renders"][name]
    data = _scene(name)
    o = ol.OracleScene(data)
    acc, img, cnt = o.render(data.camera_desc(*g["size"]))
    assert o.info() == pytest.approx(g["bvh"])
    assert {k: cnt[k] for k in g["counters"]} == g["counters"]          # (the fixture predates the per-class tallies: invariants instead)
    assert sum(cnt["n_shade_class"]) == cnt["n_shade"] and sum(cnt["n_eval_class"]) <= cnt["n_any"]
    assert crc(acc) == g["acc_crc32"] and crc(img) == g["img_crc32"]


def test_bxdf_golden():
    g = GOLD["bxdf"]
    data = _scene("mixed")
    o = ol.OracleScene(data)
    arr = {k: np.array(v, np.uint32).view(np.float32) for k, v in g["inputs"].items()}
    nrm, wo, wi = arr["normal"].reshape(-1, 3), arr["wo"].reshape(-1, 3), arr["wi"].reshape(-1, 3)
    uc, u2, uv = arr["uc"], arr["u2"].reshape(-1, 2), arr["uv"].reshape(-1, 2)
    for m, gm in g["materials"].items():
        sm = o.sampleBxdf(int(m), nrm, wo, uc, u2, uv)
        assert crc(np.concatenate([sm["ok"].astype(np.float32), sm["f"].reshape(-1), sm["wi"].reshape(-1), sm["pdf"]])) == gm["sample_crc32"]
        assert crc(o.evalBxdf(int(m), nrm, wo, wi, uv)) == gm["eval_crc32"]
        assert crc(o.pdfBxdf(int(m), nrm, wo, wi, uv)) == gm["pdf_crc32"]


def test_oracle_independent_of_threads_and_barriers():
    data = jtx.scenes.cornell()
    o = ol.OracleScene(data)
    cam = data.camera_desc(70, 45, 2, 2, 4)
    a1, i1, c1 = o.render(cam, threads=1)
    a8, i8, c8 = o.render(cam, threads=8, reference_barriers=True)
    assert np.array_equal(a1.view(np.uint32), a8.view(np.uint32)) and np.array_equal(i1, i8) and c1 == c8
    # resume: strata [0,2) then [2,4) == [0,4)
    part, _, _ = o.render(cam, sample_begin=0, sample_end=2)
    part, ip, _ = o.render(cam, sample_begin=2, sample_end=4, acc=part)
    assert np.array_equal(part.view(np.uint32), a1.view(np.uint32)) and np.array_equal(ip, i1)


def test_bxdf_physical_sanity():
    """Lambert: f = albedo/pi, pdf = |cos|/pi; sampled direction in wo's hemisphere; energy weight f*cos/pdf = albedo."""
    data = jtx.scenes.cornell()
    o = ol.OracleScene(data)
    rs = np.random.RandomState(1)
    n = 4000
    nrm = np.tile(np.array([[0, 0, 1]], np.float32), (n, 1))
    wo = rs.normal(size=(n, 3)).astype(np.float32); wo[:, 2] = np.abs(wo[:, 2]) + 0.01
    wo /= np.linalg.norm(wo, axis=1, keepdims=True).astype(np.float32)
    sm = o.sampleBxdf(0, nrm, wo, rs.uniform(0, 1, n).astype(np.float32), rs.uniform(0, 1, (n, 2)).astype(np.float32))
    ok = sm["ok"] > 0
    assert ok.mean() > 0.99
    cosi = np.abs(sm["wi"][ok, 2])
    w = sm["f"][ok] * cosi[:, None] / sm["pdf"][ok, None]
    assert np.allclose(w, 0.73, rtol=2e-5)
    assert (sm["wi"][ok, 2] > 0).all()
    assert np.allclose(sm["pdf"][ok], cosi / np.pi, rtol=2e-5)


# ------------------------------------------------------------------------------------------------
# product host logic: BVH builder and the C-ABI surface
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("maker", ["cornell", "quad", "mixed", "atrium_small", "degenerate"])
def test_host_bvh_build_equals_oracle(maker):
    if maker == "atrium_small":
        data = jtx.scenes.atrium(target_tris=6000)
    elif maker == "degenerate":
        # many coincident triangles: centroid bounds degenerate -> multi-primitive leaf (bvh.cpp:36-46)
        data = jtx.scenes.SceneData("dup")
        data.materials = [jtx.scenes.material()]
        idx, v, n, _ = jtx.scenes.quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1))
        for _ in range(5):
            data.add_mesh(idx, v, n, 0)
    else:
        data = _scene(maker)
    nodes, refs, depth = jtx.api.bvh_build_host(data)
    o = ol.OracleScene(data)
    onodes, orefs = o.bvh()
    assert nodes.tobytes() == onodes.tobytes()
    assert refs.tobytes() == orefs.tobytes()
    assert depth == o.info()["max_depth"]
    # structure: every primitive in exactly one leaf, interior boxes contain their children
    leaves = nodes[nodes["num_prims"] > 0]
    covered = np.concatenate([np.arange(l["offset"], l["offset"] + l["num_prims"]) for l in leaves]) if len(leaves) else np.array([])
    assert sorted(covered.tolist()) == list(range(data.num_triangles))
    for i, nd in enumerate(nodes):
        if nd["num_prims"] == 0:
            for c in (i + 1, nd["offset"]):
                assert (nodes[c]["pmin"] >= nd["pmin"]).all() and (nodes[c]["pmax"] <= nd["pmax"]).all()
    if maker == "degenerate":
        assert nodes["num_prims"].max() > 1


def test_bvh_build_rejects_bad_input():
    data = jtx.scenes.cornell()
    data.meshes[2]["indices"][0, 0] = 9999
    with pytest.raises(jtx.JtxMiError):
        jtx.api.bvh_build_host(data)


def test_capi_exports_every_declared_symbol():
    """libjtx_mi.so loads and exports exactly the functions include/jtx_mi.h declares."""
    lib = jtx._capi.load()
    header = open(os.path.join(ROOT, "include", "jtx_mi.h")).read()
    declared = set(re.findall(r"\b(jtx_mi_[a-z_0-9]+)\s*\(", header)) - {"jtx_mi_progress_cb"}
    assert declared == set(jtx._capi.SYMBOLS), declared ^ set(jtx._capi.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.jtx_mi_version() == 6
    assert C.sizeof(jtx._capi.BvhNode) == 32 and C.sizeof(jtx._capi.TriRef) == 8


def test_every_environment_variable_the_library_reads_is_in_the_header():
    """the tuning switches of the product library (getenv in csrc/) are documented where its interface is -- the list in
    include/jtx_mi.h is exactly what the sources read (diagnostic / test-hook reads sit behind #ifdef and are named there as such)."""
    csrc = os.path.join(ROOT, "jtx-pathtracer_amd", "csrc")
    read = set()
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".hpp", ".cpp")):
            read |= set(re.findall(r'getenv\("(JTX_[A-Z0-9_]+)"\)', open(os.path.join(csrc, f)).read()))
    header = open(os.path.join(ROOT, "include", "jtx_mi.h")).read()
    env = header[header.index("environment variables the library reads"):]
    documented = set(re.findall(r"\b(JTX_[A-Z0-9_]+)\b", env))
    assert read - documented == set(), read - documented
    assert documented - read - {"JTX_MI_H"} == set(), documented - read


def test_ctypes_structs_match_the_header(tmp_path):
    """the C-ABI's structs as a C compiler lays them out (gcc on include/jtx_mi.h: a plain C header) against the ctypes mirrors of
    _capi.py -- sizes and the offsets of the fields rounds 5 and 6 added (frame_slot, sequence_end, the per-class tallies, the spare-set bytes; max_record_mb, frame_slot_bytes)"""
    import subprocess
    src = tmp_path / "sizes.c"
    src.write_text('''#include <stdio.h>
#include <stddef.h>
#include "jtx_mi.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %d %zu %zu\\n", sizeof(jtx_mi_render_opts), offsetof(jtx_mi_render_opts, frame_slot),
           offsetof(jtx_mi_render_opts, sequence_end), sizeof(jtx_mi_counters), offsetof(jtx_mi_counters, n_shade_class),
           offsetof(jtx_mi_counters, n_eval_class), sizeof(jtx_mi_scene_info), offsetof(jtx_mi_scene_info, wide_bytes64),
           offsetof(jtx_mi_scene_info, rebuild_spare_bytes), sizeof(jtx_mi_camera_desc), JTX_MI_FRAME_SLOTS,
           offsetof(jtx_mi_render_opts, max_record_mb), offsetof(jtx_mi_scene_info, frame_slot_bytes));
    return 0;
}
''')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    k = jtx._capi
    want = [C.sizeof(k.RenderOpts), k.RenderOpts.frame_slot.offset, k.RenderOpts.sequence_end.offset, C.sizeof(k.Counters),
            k.Counters.n_shade_class.offset, k.Counters.n_eval_class.offset, C.sizeof(k.SceneInfo), k.SceneInfo.wide_bytes64.offset,
            k.SceneInfo.rebuild_spare_bytes.offset, C.sizeof(k.CameraDesc), jtx.distributed.FRAME_SLOTS,
            k.RenderOpts.max_record_mb.offset, k.SceneInfo.frame_slot_bytes.offset]
    assert got == want


def test_no_cpu_fallback_in_product():
    """Without a GPU every compute entry point must fail loudly, never fall back to the CPU."""
    lib = jtx._capi.load()
    n = C.c_int32(0)
    if lib.jtx_mi_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    sc = jtx.Scene(jtx.scenes.cornell())
    with pytest.raises(jtx.JtxMiError):
        sc.buildBVH()
    # and nothing in the product package includes, imports, links or loads anything under oracle/
    pkg = os.path.join(ROOT, "jtx-pathtracer_amd")
    bad = re.compile(r"jtx_oracle|oracle_lib|oracle/|libjtx_oracle|import\s+oracle|ora_[a-z_]+\(")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not bad.search(text), f"{f} references the oracle"
    assert not bad.search(open(os.path.join(ROOT, "include", "jtx_mi.h")).read())


def test_scene_generators_are_deterministic_and_sized():
    a, b = jtx.scenes.atrium(target_tris=20000), jtx.scenes.atrium(target_tris=20000)
    assert a.num_triangles == b.num_triangles
    for ma, mb in zip(a.meshes, b.meshes):
        assert ma["vertices"].tobytes() == mb["vertices"].tobytes()
    assert 10000 < a.num_triangles < 40000
    m = jtx.scenes.mixed(sphere_res=(16, 8))
    assert {x["type"] for x in m.materials} == {0, 1, 2, 3} and len(m.textures) == 2 and len(m.lights) == 2
    assert jtx.scenes.cornell().num_triangles == 32


# ------------------------------------------------------------------------------------------------
# OBJ ingestion and the restated Cornell data
# ------------------------------------------------------------------------------------------------
def test_obj_round_trip(tmp_path):
    ref = jtx.scenes.cornell()
    path = str(tmp_path / "cb.obj")
    jtx.scenes.write_obj(ref, path)
    names = {"left_wall": ref.materials[1], "right_wall": ref.materials[2]}
    got = jtx.scenes.load_obj(path, default_material=ref.materials[0], materials_by_name=names)
    assert len(got.meshes) == len(ref.meshes) == 8 and got.num_triangles == 32
    for a, b in zip(got.meshes, ref.meshes):
        assert a["name"] == b["name"]
        assert a["vertices"].tobytes() == b["vertices"].tobytes()
        assert a["normals"].tobytes() == b["normals"].tobytes()
        assert a["indices"].tobytes() == b["indices"].tobytes()
        assert got.materials[a["material"]]["albedo"] == ref.materials[b["material"]]["albedo"]
    # same scene => same BVH
    got.lights, got.sky, got.camera = ref.lights, ref.sky, ref.camera
    n1, r1, _ = jtx.api.bvh_build_host(got)
    n2, r2, _ = jtx.api.bvh_build_host(ref)
    assert n1.tobytes() == n2.tobytes() and r1.tobytes() == r2.tobytes()


def test_obj_reader_semantics(tmp_path):
    """quads are fanned, missing normals are generated flat, v is flipped, negative indices work."""
    p = tmp_path / "q.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 0.25\n"
                 "o quad\nf 1/1 2/2 3/3 4/4\no tri\nf -4 -3 -2\n")
    s = jtx.scenes.load_obj(str(p))
    assert [len(m["indices"]) for m in s.meshes] == [2, 1]
    q, t = s.meshes
    assert q["vertices"].shape == (6, 3) and np.allclose(q["normals"], [0, 0, 1])
    assert np.allclose(q["uvs"][5], [0, 0.75])                  # (0, 0.25) flipped
    assert t["uvs"] is None and np.allclose(t["normals"], [0, 0, 1])


@pytest.mark.skipif(not os.path.exists("/root/reference/src/assets/scenes/cornell_box.obj"),
                    reason="reference assets are only present in the build container")
def test_restated_cornell_equals_reference_asset():
    """scenes.cornell() restates the 32 triangles of the reference's cornell_box.obj as constants; where the
    asset is present, prove the restatement: same objects, faces, positions and normals."""
    got = jtx.scenes.load_obj("/root/reference/src/assets/scenes/cornell_box.obj")
    ref = jtx.scenes.cornell()
    assert [m["name"] for m in got.meshes] == [m["name"] for m in ref.meshes]
    for a, b in zip(got.meshes, ref.meshes):
        assert a["vertices"].tobytes() == b["vertices"].tobytes(), a["name"]
        assert a["normals"].tobytes() == b["normals"].tobytes(), a["name"]
        assert a["indices"].tobytes() == b["indices"].tobytes()


# ------------------------------------------------------------------------------------------------
# kernel-level known answers (AABB::hit pairs, ray-triangle grazers, the Cornell BVH, per-sample radiance)
# ------------------------------------------------------------------------------------------------
def _golden_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_kernel_known_answers_oracle():
    mg = _golden_module()
    scn = {"cornell": jtx.scenes.cornell()}
    assert mg.kernel_vectors(scn) == GOLD["kernels"]
    # the AABB vector once more with the independent numpy restatement of aabb.hpp:66-81
    k = mg.kernel_inputs()
    hits = np.array([np_aabb_hit(k["lo"][i], k["hi"][i], k["o"][i], k["d"][i], 0.001, float(k["t1"][i])) for i in range(len(k["lo"]))], np.uint8)
    assert mg.crc(hits) == GOLD["kernels"]["aabb_hit_crc32"] and int(hits.sum()) == GOLD["kernels"]["aabb_hit_count"]
    assert 1000 < hits.sum() < 3000
    # the ray set really contains grazers: rays through triangle vertices / edge midpoints that still hit
    assert GOLD["kernels"]["cornell_closest"]["hits"] > 0.9 * GOLD["kernels"]["cornell_closest"]["rays"]


# ---- the oracle's restatements of integrate / integrateBasic / ThinDielectricBxDF (no GPU) ----
def test_thin_dielectric_sample_properties():
    """ThinDielectricBxDF::sample (dielectric.hpp:173-199) on the oracle: two specular lobes, reflection (-x,-y,z) with
    probability R' and straight transmission -wo with T' = 1 - R', R' = R + T^2 R / (1 - R^2); f |cos| / pdf == 1."""
    from jtx_pathtracer_amd import scenes
    s = scenes.quad_scene()
    s.materials = [scenes.material(scenes.THIN_DIELECTRIC, ior=(1.5, 1.5, 1.5))]
    for m in s.meshes:
        m["material"] = 0
    o = ol.OracleScene(s)
    rs = np.random.RandomState(5)
    n = 4000
    nrm = np.tile(np.array([0, 0, 1], np.float32), (n, 1))
    wo = rs.normal(size=(n, 3)).astype(np.float32); wo /= np.linalg.norm(wo, axis=1, keepdims=True); wo[:, 2] = np.abs(wo[:, 2]) + 0.05
    wo /= np.linalg.norm(wo, axis=1, keepdims=True).astype(np.float32)
    uc = rs.uniform(0, 1, n).astype(np.float32); u2 = rs.uniform(0, 1, (n, 2)).astype(np.float32)
    r = o.sampleBxdf(0, nrm, wo, uc, u2)
    assert r["ok"].all()
    refl = np.isclose(r["wi"][:, 2], wo[:, 2], atol=1e-5)
    trans = np.isclose(r["wi"], -wo, atol=1e-5).all(axis=1)
    assert (refl ^ trans).all() and refl.any() and trans.any()
    w = r["f"][:, 0] * np.abs(r["wi"][:, 2]) / r["pdf"]
    assert np.allclose(w, 1.0, atol=2e-6)                               # energy-preserving pane
    c = wo[:, 2].astype(np.float64)
    ct = np.sqrt(1 - (1 - c * c) / 2.25)
    R = 0.5 * (((1.5 * c - ct) / (1.5 * c + ct)) ** 2 + ((c - 1.5 * ct) / (c + 1.5 * ct)) ** 2)
    Rp = R + (1 - R) ** 2 * R / (1 - R * R)
    assert np.allclose(np.where(refl, r["pdf"], 1 - r["pdf"]), Rp, atol=1e-5)
    assert ((uc < Rp - 1e-5) <= refl).all() and ((uc > Rp + 1e-5) <= trans).all()


def test_oracle_alternate_integrators_basic_properties():
    """integrateBasic sees emitters and never traces a shadow ray; integrate needs a specular bounce (or the camera ray)
    to collect emission; integrateMIS collects none (integrator.cpp:189-190)."""
    from jtx_pathtracer_amd import scenes
    s = scenes.emissive(sphere_res=(8, 4))
    o = ol.OracleScene(s)
    cam = s.camera_desc(64, 40, 2, 2, 4)
    a0, _, c0 = o.render(cam, path_integrator=0)
    a1, _, c1 = o.render(cam, path_integrator=1)
    a2, _, c2 = o.render(cam, path_integrator=2)
    assert c2["n_any"] == 0 and c0["n_any"] > 0 and 0 < c1["n_any"] <= c0["n_any"] + c1["n_shade"]
    assert c0["n_camera"] == c1["n_camera"] == c2["n_camera"] == 64 * 40 * 4
    # remove every emitter: integrateBasic loses all light but the sky; integrateMIS does not change at all
    for m in s.materials:
        m["emission"] = (0, 0, 0)
    o2 = ol.OracleScene(s)
    b0, _, _ = o2.render(cam, path_integrator=0)
    b2, _, _ = o2.render(cam, path_integrator=2)
    assert np.array_equal(a0.view(np.uint32), b0.view(np.uint32))
    assert b2.sum() < 0.7 * a2.sum()                                   # what is left is the sky seen directly and by bounces
    # the camera ray counts as a specular bounce: pixels that look straight at the emissive quad are lit under integrate
    assert a1.max() > 0 and np.isfinite(a1).all() and np.isfinite(a2).all()


def test_host_bvh_build_says_when_the_reference_would_not_terminate():
    """boxes whose surface area overflows fp32 make every SAH cost inf / NaN: no cost is < INF, minBucket stays -1 (bvh.cpp:96-101),
    std::partition leaves one side empty and the reference's buildTree recurses on the same span for ever (bvh.cpp:126-127).  The host
    builder used to follow it until std::bad_alloc; it names the situation now."""
    from jtx_pathtracer_amd import api, scenes as sc_
    s = sc_.SceneData("overflow")
    s.materials = [sc_.material(sc_.DIFFUSE, (0.7, 0.6, 0.5))]
    v = []
    for k in range(140):
        e = np.float32(1.5) ** k * np.float32(0.01)
        v += [(0, 0, -k * 1e-3), (e, 0, -k * 1e-3), (0, e, -k * 1e-3)]
    v = np.array(v, np.float32)
    s.add_mesh(np.arange(len(v), dtype=np.int32).reshape(-1, 3), v, np.tile(np.array([[0, 0, 1]], np.float32), (len(v), 1)), 0)
    with pytest.raises(Exception, match="does not terminate"):
        api.bvh_build_host(s)


def test_film_buffers_of_the_python_mirror_have_pages_of_their_own():
    """api._film_array: the arrays the cameras page-lock are zeroed, writable, page-aligned private mappings (not numpy-allocator blocks
    that share their first and last page with other heap objects) and survive the camera object"""
    import gc
    import numpy as np
    from jtx_pathtracer_amd import api
    a = api._film_array((120, 200, 3), np.float32)
    assert a.shape == (120, 200, 3) and a.dtype == np.float32 and a.flags.writeable and a.flags.c_contiguous
    assert a.ctypes.data % 4096 == 0 and not a.any()
    a[...] = 2.0
    assert api._film_array((0, 0, 3), np.uint8).shape == (0, 0, 3)
    cam = api.StaticCamera(64, 48, dict(center=(0, 0, 0), target=(0, 0, 1), up=(0, 1, 0), yfov=40.0, defocus_angle=0.0, focus_distance=1.0), 2, 2, 3)
    img, acc = cam.img_, cam.acc_
    assert img.ctypes.data % 4096 == 0 and acc.ctypes.data % 4096 == 0 and img.shape == (48, 64, 3) and acc.dtype == np.float32
    del cam; gc.collect()
    acc[...] = 1.0; img[...] = 7                                                  # the mappings live as long as the arrays
    assert float(acc.sum()) == 48 * 64 * 3 and int(img[0, 0, 0]) == 7 and float(a[5, 5, 1]) == 2.0
