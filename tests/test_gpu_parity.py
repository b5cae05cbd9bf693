"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle, bit for bit.

Tolerance: NONE.  All arithmetic on the path is IEEE fp32 with one rounding per operation on both
sides (device code is built -ffp-contract=off, sin/cos are the deterministic polynomials of
DESIGN.md), so radiance, accumulation buffer, RGB8 image and ray counters must be identical.
"""
import numpy as np
import pytest

import os

import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def atrium_full(gpu):
    data = gpu.scenes.atrium()                            # C3 / C4 scene: ~262 k triangles
    sc = gpu.Scene(data); sc.buildBVH()
    yield data, sc, ol.OracleScene(data)
    sc.destroy()


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_same_f32(a, b, what):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, what
    # NaNs compare by bit pattern class: both NaN is fine
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        bad = np.argwhere(~same)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {a.size} floats differ; first at {i}: gpu={a[i]!r} oracle={b[i]!r}")


def test_rng_streams(gpu):
    for seed in [(0, 0, 1), (1, 2, 3), (511, 17, 16), (1079, 1919, 64), (4000000000, 7, 9)]:
        gu, gf = gpu.api.rng_stream(*seed, 64)
        ou, of = ol.rng_stream(*seed, 64)
        assert (gu == ou).all()
        assert_same_f32(gf, of, "rng floats")


def test_sincos_bitexact(gpu):
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.uniform(-0.8, 2 * np.pi + 0.1, 200000), rs.uniform(-16, 16, 50000),
                        np.array([0.0, -0.0, np.pi / 4, np.pi / 2, np.pi, 2 * np.pi, 1e-8, -1e-8])]).astype(np.float32)
    gs, gc = gpu.api.sincos(x)
    os_, oc = ol.sincos(x)
    assert_same_f32(gs, os_, "sin")
    assert_same_f32(gc, oc, "cos")


def _grid_samples(w, h, spp, n, seed):
    rs = np.random.RandomState(seed)
    return rs.randint(0, h, n).astype(np.int32), rs.randint(0, w, n).astype(np.int32), rs.randint(0, spp, n).astype(np.int32)


@pytest.mark.parametrize("defocus", [0.0, 2.5])
def test_camera_rays(gpu, defocus):
    data = gpu.scenes.cornell()
    data.camera["defocus_angle"] = defocus
    data.camera["focus_distance"] = 1.0 if defocus == 0 else 800.0
    cam = data.camera_desc(640, 360, 8, 4, 5)
    row, col, smp = _grid_samples(640, 360, 32, 20000, 1)
    go, gd = gpu.api.camera_rays(cam, row, col, smp)
    oo, od = ol.camera_rays(cam, row, col, smp)
    assert_same_f32(go, oo, "ray origin")
    assert_same_f32(gd, od, "ray dir")


def _rays_for(data, osc, n, seed):
    """camera rays + secondary-like rays from hit points + axis-parallel / grazing rays."""
    cam = data.camera_desc(256, 256, 2, 2, 4)
    row, col, smp = _grid_samples(256, 256, 4, n, seed)
    o, d = ol.camera_rays(cam, row, col, smp)
    h = osc.closestHit(o, d)
    rs = np.random.RandomState(seed + 1)
    dirs = rs.normal(size=(n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True).astype(np.float32)
    o2 = np.where(h["hit"][:, None] > 0, h["point"] + 1e-3 * dirs, o).astype(np.float32)
    axis = np.zeros((64, 3), np.float32)
    axis[np.arange(64), np.arange(64) % 3] = np.where(np.arange(64) % 2, 1.0, -1.0)
    o3 = np.tile(np.array([[278.0, 273.0, 200.0]], np.float32), (64, 1))
    return (np.concatenate([o, o2, o3]), np.concatenate([d, dirs, axis]))


@pytest.mark.parametrize("pair", ["cornell_pair", "mixed_pair"])
def test_closest_hit(pair, request):
    data, sc, osc = request.getfixturevalue(pair)
    o, d = _rays_for(data, osc, 30000, 3)
    g = sc.closestHit(o, d)
    r = osc.closestHit(o, d)
    assert (g["hit"] == r["hit"]).all()
    assert (g["prim"] == r["prim"]).all()
    for k in ("t", "b1", "b2", "point", "normal", "uv"):
        assert_same_f32(g[k], r[k], k)
    assert g["hit"].mean() > 0.5


@pytest.mark.parametrize("pair", ["cornell_pair", "mixed_pair"])
def test_any_hit(pair, request):
    data, sc, osc = request.getfixturevalue(pair)
    o, d = _rays_for(data, osc, 30000, 5)
    rs = np.random.RandomState(9)
    tmax = rs.uniform(1.0, 900.0, len(o)).astype(np.float32)
    g = sc.anyHit(o, d, 0.0, tmax)
    r = osc.anyHit(o, d, 0.0, tmax)
    assert (g == r).all()
    assert 0.05 < g.mean() < 0.95


def _bxdf_inputs(n, seed):
    rs = np.random.RandomState(seed)
    nrm = rs.normal(size=(n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
    nrm[: n // 4] *= rs.uniform(0.9, 1.1, (n // 4, 1)).astype(np.float32)      # non-unit normals (quirk Q6)
    wo = rs.normal(size=(n, 3)).astype(np.float32)
    wo[n // 2:] /= np.linalg.norm(wo[n // 2:], axis=1, keepdims=True).astype(np.float32)   # half unit, half not (Q5)
    wi = rs.normal(size=(n, 3)).astype(np.float32)
    wi /= np.linalg.norm(wi, axis=1, keepdims=True).astype(np.float32)
    # grazing: wo nearly perpendicular to the normal
    g = slice(0, n // 10)
    wo[g] = np.cross(nrm[g], wi[g]) + 1e-4 * nrm[g]
    uc = rs.uniform(0, 1, n).astype(np.float32)
    u2 = rs.uniform(0, 1, (n, 2)).astype(np.float32)
    uv = rs.uniform(-2, 3, (n, 2)).astype(np.float32)
    return nrm, wo, wi, uc, u2, uv


def test_bxdf_all_materials(mixed_pair):
    data, sc, osc = mixed_pair
    nrm, wo, wi, uc, u2, uv = _bxdf_inputs(20000, 11)
    assert len(data.materials) >= 11
    for m in range(len(data.materials)):
        gs = sc.sampleBxdf(m, nrm, wo, uc, u2, uv)
        rs_ = osc.sampleBxdf(m, nrm, wo, uc, u2, uv)
        assert (gs["ok"] == rs_["ok"]).all(), f"material {m} sample ok"
        for k in ("f", "wi", "pdf"):
            assert_same_f32(gs[k], rs_[k], f"material {m} sample {k}")
        assert_same_f32(sc.evalBxdf(m, nrm, wo, wi, uv), osc.evalBxdf(m, nrm, wo, wi, uv), f"material {m} eval")
        assert_same_f32(sc.pdfBxdf(m, nrm, wo, wi, uv), osc.pdfBxdf(m, nrm, wo, wi, uv), f"material {m} pdf")
        if data.materials[m]["type"] != 1 or data.materials[m]["alpha_x"] > 0:
            assert gs["ok"].mean() > 0.3, f"material {m}: too few successful samples"


@pytest.mark.parametrize("pair,depth", [("cornell_pair", 4), ("cornell_pair", 8), ("mixed_pair", 8)])
def test_radiance_samples(pair, depth, request):
    data, sc, osc = request.getfixturevalue(pair)
    cam = data.camera_desc(320, 200, 4, 4, depth)
    row, col, smp = _grid_samples(320, 200, 16, 40000, 21)
    g = gpu_rgb = request.getfixturevalue("gpu").api.radiance_samples(sc, cam, row, col, smp)
    r = osc.radiance_samples(cam, row, col, smp)
    assert_same_f32(g, r, "radiance")
    assert gpu_rgb.max() > 0.1


def _render_both(gpu, data, sc, osc, w, h, xs, ys, depth, **kw):
    cam_g = gpu.StaticCamera(w, h, data.camera, xs, ys, depth)
    cam_g.render(sc, count_rays=True, **kw)
    acc, img, cnt = osc.render(data.camera_desc(w, h, xs, ys, depth))
    return cam_g, acc, img, cnt


@pytest.mark.parametrize("integrator", [1, 2])
def test_render_config1_cornell_512(gpu, cornell_pair, integrator):
    """BASELINE config 1: Cornell 512x512, 16 spp (4x4), depth 4 -- whole frame, bit-exact, both integrators."""
    data, sc, osc = cornell_pair
    cam_g, acc, img, cnt = _render_both(gpu, data, sc, osc, 512, 512, 4, 4, 4, integrator=integrator)
    assert_same_f32(cam_g.acc_, acc, "accumulation buffer")
    assert (cam_g.img_ == img).all()
    assert cam_g.counters == cnt
    assert cnt["n_camera"] == 512 * 512 * 16


@pytest.mark.parametrize("integrator", [1, 2])
def test_render_mixed_small(gpu, mixed_pair, integrator):
    data, sc, osc = mixed_pair
    cam_g, acc, img, cnt = _render_both(gpu, data, sc, osc, 200, 120, 2, 2, 8, integrator=integrator)
    assert_same_f32(cam_g.acc_, acc, "accumulation buffer")
    assert (cam_g.img_ == img).all()
    assert cam_g.counters == cnt


@pytest.mark.parametrize("integrator", [1, 2])
def test_render_ragged_sizes(gpu, cornell_pair, integrator):
    """widths/heights that are not multiples of the 8x8 wave block or the 32x32 tile; 1x1 image."""
    data, sc, osc = cornell_pair
    for (w, h) in [(1, 1), (33, 7), (70, 45)]:
        cam_g, acc, img, cnt = _render_both(gpu, data, sc, osc, w, h, 2, 1, 3, integrator=integrator)
        assert_same_f32(cam_g.acc_, acc, f"acc {w}x{h}")
        assert (cam_g.img_ == img).all()
        assert cam_g.counters == cnt


def test_render_resume_and_progress(gpu, cornell_pair):
    """rendering strata [0,3) then [3,8) equals [0,8); the progress callback sees every pass."""
    data, sc, osc = cornell_pair
    full = gpu.StaticCamera(96, 64, data.camera, 4, 2, 4); full.render(sc)
    part = gpu.StaticCamera(96, 64, data.camera, 4, 2, 4)
    part.render(sc, sample_begin=0, sample_end=3)
    part.render(sc, sample_begin=3, sample_end=8)
    assert_same_f32(part.acc_, full.acc_, "resumed acc")
    assert (part.img_ == full.img_).all()
    seen = []
    prog = gpu.StaticCamera(96, 64, data.camera, 4, 2, 4)
    prog.render(sc, progress=lambda c, t: seen.append((c, t)))
    assert seen == [(i, 8) for i in range(1, 9)]
    assert_same_f32(prog.acc_, full.acc_, "progressive acc")
    # cancellation: terminateRender() stops after the current pass
    stop = gpu.StaticCamera(96, 64, data.camera, 4, 2, 4)
    stop.render(sc, progress=lambda c, t: stop.terminateRender() if c == 2 else None)
    # all passes of the frame are ONE launch (round 6: jtx_mi.h, jtx_mi_render), which does not wait for the callback: on a frame this
    # small the later passes may have run to their end before the waves look at the flag -- then their strata ARE in the film and
    # count; never a partial pass, never a pass without the one before it
    n = stop.currentSample_
    assert 2 <= n <= 8
    two = gpu.StaticCamera(96, 64, data.camera, 4, 2, 4); two.render(sc, sample_begin=0, sample_end=n)
    assert_same_f32(stop.acc_, two.acc_, "film after a cancellation = the completed passes, nothing of the abandoned one")
    assert (stop.img_ == two.img_).all()


def test_pageable_caller_buffers_take_the_staged_path(gpu, cornell_pair, monkeypatch):
    """Film buffers the caller did NOT page-lock (a C++ host's plain vectors; JTX_PIN_CAMERA_BUFFERS=0 here): delivery and the upload of a
    resumed accumulation go through the library's own page-locked staging (no pageable pointer reaches a HIP copy: DESIGN.md section
    10, "found by the project's own test runs") -- same film, bit for bit, incl. a frame larger than one 8 MB staging chunk."""
    data, sc, osc = cornell_pair
    pinned = gpu.StaticCamera(96, 64, data.camera, 4, 2, 4); pinned.render(sc)
    assert pinned._pinned and pinned.acc_.ctypes.data % 4096 == 0             # the mirror's own buffers: page-aligned private mappings, page-locked
    monkeypatch.setenv("JTX_PIN_CAMERA_BUFFERS", "0")
    plain = gpu.StaticCamera(96, 64, data.camera, 4, 2, 4)
    plain.render(sc, sample_begin=0, sample_end=3)
    plain.render(sc, sample_begin=3, sample_end=8)                             # the resumed accumulation is uploaded from pageable memory
    assert plain._pinned == []
    assert_same_f32(plain.acc_, pinned.acc_, "film through the staging buffers"); assert (plain.img_ == pinned.img_).all()
    # ... and the previews of a progressive launch: copied to the library's page-locked buffer, then into the caller's pageable one
    prog = gpu.StaticCamera(96, 64, data.camera, 4, 2, 4); prog.samplesPerPass_ = 2
    seen, last_preview = [], []
    def tick(c, t):
        seen.append(c)
        if c == t: last_preview.append(prog.img_.copy())
    prog.render(sc, progress=tick)
    assert prog._pinned == [] and seen == [2, 4, 6, 8]
    assert_same_f32(prog.acc_, pinned.acc_, "progressive film through the staging buffers"); assert (prog.img_ == pinned.img_).all()
    assert len(last_preview) == 1 and (last_preview[0] == pinned.img_).all()     # the preview of the last pass is the finished image
    big_p =gpu.StaticCamera(1200, 800, data.camera, 1, 1, 3)                  # 11.5 MB of accumulation: two staging chunks
    big_u = gpu.StaticCamera(1200, 800, data.camera, 1, 1, 3)
    big_u.render(sc)
    monkeypatch.setenv("JTX_PIN_CAMERA_BUFFERS", "1")
    big_p.render(sc)
    assert big_u._pinned == [] and big_p._pinned
    assert_same_f32(big_u.acc_, big_p.acc_, "large film through the staging buffers"); assert (big_u.img_ == big_p.img_).all()


def test_cancel_reaches_the_pass_in_flight(gpu, atrium_full):
    """Camera::terminateRender from another thread (the UI thread, display.cpp:899-910) while ONE long pass is on the
    GPU (the reference polls stopRender_ per pixel, camera.cpp:84-98): the persistent waves stop fetching chunks, the
    call returns early with the film untouched, and the next render of the same scene is complete and correct."""
    import threading, time
    data, sc, osc = atrium_full
    cam = gpu.StaticCamera(1920, 1080, data.camera, 8, 8, 8)
    t0 = time.perf_counter(); cam.render(sc); t_full = time.perf_counter() - t0
    ref_acc = cam.acc_.copy(); ref_img = cam.img_.copy()
    assert cam.currentSample_ == 64
    timer = threading.Timer(0.03, cam.terminateRender)
    t0 = time.perf_counter(); timer.start(); cam.render(sc); t_cancel = time.perf_counter() - t0
    timer.join()
    assert cam.currentSample_ == 0 and not cam.acc_.any() and not cam.img_.any(), "an abandoned pass must leave no trace"
    assert t_cancel < 0.6 * t_full, (t_cancel, t_full)
    cam.render(sc)                                                    # stopRender_ = false again (camera.cpp:48)
    assert cam.currentSample_ == 64
    assert_same_f32(cam.acc_, ref_acc, "render after a cancelled one"); assert (cam.img_ == ref_img).all()


def test_dynamic_camera_on_a_second_device(gpu, cornell_pair):
    """Every scene-taking entry point switches to the scene's device: a DynamicCamera worker thread (which never called
    hipSetDevice) renders a scene that lives on device 1.  Needs two visible devices."""
    import ctypes as C
    lib = gpu._capi.load()
    n = C.c_int32(0); gpu._capi.check(lib.jtx_mi_device_count(C.byref(n)))
    if n.value < 2:
        pytest.skip("one visible device")
    data, sc0, osc = cornell_pair
    gpu._capi.check(lib.jtx_mi_set_device(1))
    try:
        sc1 = gpu.Scene(data); sc1.buildBVH()
    finally:
        gpu._capi.check(lib.jtx_mi_set_device(0))
    ref = gpu.StaticCamera(96, 64, data.camera, 2, 2, 4); ref.render(sc0)
    dyn = gpu.DynamicCamera(96, 64, data.camera, 2, 2, 4, samplesPerPass=1)
    dyn.render(sc1); assert dyn.wait(60)
    dyn.stopRender()
    assert_same_f32(dyn.acc_, ref.acc_, "DynamicCamera on device 1"); assert (dyn.img_ == ref.img_).all()
    sc1.destroy()


@pytest.mark.parametrize("integrator", [1, 2])
def test_render_tile_sharding(gpu, cornell_pair, integrator):
    """pixel-tile shards of 3 ranks are disjoint, zero elsewhere, and sum to the 1-GPU frame exactly."""
    data, sc, osc = cornell_pair
    full = gpu.StaticCamera(200, 100, data.camera, 2, 2, 4); full.render(sc, integrator=integrator)
    total = np.zeros_like(full.acc_)
    cover = np.zeros(full.acc_.shape[:2], np.int32)
    for r in range(3):
        c = gpu.StaticCamera(200, 100, data.camera, 2, 2, 4)
        c.render(sc, tile_rank=r, tile_world=3, integrator=integrator)
        total += c.acc_
        cover += (gpu.distributed.tile_owner_mask(200, 100, r, 3)).astype(np.int32)
        assert (c.acc_[~gpu.distributed.tile_owner_mask(200, 100, r, 3)] == 0).all()
    assert (cover == 1).all()
    assert_same_f32(total, full.acc_, "sum of shards")


def test_empty_and_degenerate_scenes(gpu):
    """no triangles at all (every ray misses -> sky), and a scene without lights."""
    s = gpu.scenes.SceneData("empty")
    s.materials = [gpu.scenes.material()]
    s.sky = (0.25, 0.5, 0.75)
    sc = gpu.Scene(s); sc.buildBVH()
    cam = gpu.StaticCamera(16, 8, s.camera, 1, 1, 3); cam.render(sc)
    assert_same_f32(cam.acc_, np.broadcast_to(np.array(s.sky, np.float32), (8, 16, 3)), "sky only")
    q = gpu.scenes.quad_scene()
    sq = gpu.Scene(q); sq.buildBVH()
    oq = ol.OracleScene(q)
    cg = gpu.StaticCamera(64, 64, q.camera, 2, 2, 4); cg.render(sq)
    acc, img, _ = oq.render(q.camera_desc(64, 64, 2, 2, 4))
    assert_same_f32(cg.acc_, acc, "quad scene acc")
    assert (cg.img_ == img).all()


def test_errors_are_reported(gpu):
    s = gpu.scenes.cornell()
    s.meshes[0]["material"] = 99
    with pytest.raises(gpu.JtxMiError):
        gpu.Scene(s).buildBVH()


def test_cpp_host_mirror_reproduces_reference_probe(gpu, tmp_path):
    """The C++ Scene/StaticCamera mirror (jtx-pathtracer_amd/host/jtx_host_api.hpp) driven like the reference's
    main.cpp: createMeshScene quad, 64x64, 2x2 spp, depth 4.  The reference's own sources give the RGB8
    FNV-1a hash 1af9ba89 for this set-up."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_api_demo")
    libdir = os.path.join(root, "jtx-pathtracer_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(root, "tests", "cpp", "host_api_demo.cpp"),
                    "-L", libdir, "-ljtx_mi", "-lpthread", "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    kv = dict(zip(out[0::2], out[1::2]))
    assert kv["hash"] == "1af9ba89"
    assert kv["samples"] == "4" and kv["hit"] == "1" and kv["shadow"] == "0"
    assert abs(float(kv["t"]) - 9.0) < 1e-5 and abs(float(kv["radius"]) - 2 ** 0.5) < 1e-4
    assert kv["boundsmin"] == "-1,-1,-1" and kv["boundsmax"] == "1,1,-1"    # Scene::bounds() (scene.hpp:71-74): the quad's box
    assert kv["rebuildhash"] == "1af9ba89"                                  # Scene::rebuildBVHOnDevice (jtx_mi_scene_rebuild)
    # DynamicCamera: same image as the static render; a restarted render equals a static one from the new camera
    assert kv["dynhash"] == "1af9ba89" and kv["dynsamples"] == "4" and kv["restart_same"] == "1"
    assert kv["multihash"] == "1af9ba89" and kv["multisamples"] == "4"      # 3 shards through jtx_mi_multi_render


@pytest.mark.parametrize("integrator", [1, 2])
def test_render_atrium_small(gpu, integrator):
    """config C3's scene family (deep BVH in HBM, DISTANT light + sky) at a size the oracle renders in seconds."""
    data = gpu.scenes.atrium(target_tris=20000)
    sc = gpu.Scene(data); sc.buildBVH()
    assert not sc.info()["lds_resident"]
    osc = ol.OracleScene(data)
    cam_g, acc, img, cnt = _render_both(gpu, data, sc, osc, 160, 90, 2, 2, 8, integrator=integrator)
    assert_same_f32(cam_g.acc_, acc, "accumulation buffer")
    assert (cam_g.img_ == img).all()
    assert cam_g.counters == cnt


def _frame_on_device(gpu, sc, cam, integrator, **kw):
    import torch
    H, W = cam.height, cam.width
    dev = torch.device("cuda", 0)
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    img = torch.zeros(H * W * 3, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())      # the fills above ran on the default stream
    with torch.cuda.stream(st):
        gpu.distributed.render_shard(sc, cam, kw.get("rank", 0), kw.get("world", 1), acc, img, stream=st.cuda_stream,
                                     integrator=integrator, count_rays=kw.get("count", False), max_record_mb=kw.get("max_record_mb", 0))
    torch.cuda.synchronize()
    cnt = None
    if kw.get("count"):
        import ctypes as C
        c = gpu._capi.Counters()
        gpu._capi.check(gpu._capi.load().jtx_mi_get_counters(sc.handle, C.byref(c)))
        cnt = c.as_dict()
    return acc.cpu().numpy().reshape(H, W, 3), img.cpu().numpy().reshape(H, W, 3), cnt


def test_full_size_config2_properties(gpu, cornell_pair):
    """BASELINE config 2 at full size (1920x1080, 64 spp, depth 8), through size-independent properties:
    two independent integrators agree bit for bit; counter invariants; 8 tile shards sum to the frame;
    and 192 pixels picked at random equal the oracle's in-order sum of their 64 per-sample radiances."""
    data, sc, osc = cornell_pair
    cam = data.camera_desc(1920, 1080, 8, 8, 8)
    acc1, img1, cnt1 = _frame_on_device(gpu, sc, cam, 1, count=True)
    acc3, img3, cnt3 = _frame_on_device(gpu, sc, cam, 2, count=True)
    assert np.array_equal(acc1.view(np.uint32), acc3.view(np.uint32)) and np.array_equal(img1, img3)
    assert cnt1 == cnt3
    assert cnt1["n_camera"] == 1920 * 1080 * 64
    assert cnt1["n_shade"] == cnt1["n_any"]                    # one light: one shadow ray per shading event
    assert cnt1["n_camera"] <= cnt1["n_closest"] <= cnt1["n_camera"] * 9
    assert abs((cnt1["n_closest"] + cnt1["n_any"]) / cnt1["n_camera"] - 5.49) < 0.01
    total = np.zeros_like(acc1)
    for r in range(8):
        a, _, _ = _frame_on_device(gpu, sc, cam, 1, rank=r, world=8)
        total += a
    assert np.array_equal(total.view(np.uint32), acc1.view(np.uint32))
    rs = np.random.RandomState(4)
    rows, cols = rs.randint(0, 1080, 192), rs.randint(0, 1920, 192)
    rr = np.repeat(rows, 64).astype(np.int32); cc = np.repeat(cols, 64).astype(np.int32)
    ss = np.tile(np.arange(64, dtype=np.int32), 192)
    rad = osc.radiance_samples(cam, rr, cc, ss).reshape(192, 64, 3)
    expect = np.zeros((192, 3), np.float32)
    for s in range(64):
        expect = expect + rad[:, s]
    assert np.array_equal(acc1[rows, cols].view(np.uint32), expect.view(np.uint32))
    expect_img = (255.999 * np.clip(np.sqrt(np.maximum(expect / np.float32(64), 0)), 0, 0.999).astype(np.float32)).astype(np.float32)
    assert np.abs(img1[rows, cols].astype(np.int32) - expect_img.astype(np.int32)).max() <= 1


def test_full_size_config3_properties(gpu):
    """BASELINE config 3 (atrium ~262 k triangles, 1920x1080; 16 of the 64 strata to bound the run): the
    pixel-persistent and the wavefront integrators agree bit for bit; pixels checked against the oracle."""
    data = gpu.scenes.atrium()
    sc = gpu.Scene(data); sc.buildBVH()
    cam = data.camera_desc(1920, 1080, 4, 4, 8)
    acc1, img1, cnt1 = _frame_on_device(gpu, sc, cam, 1, count=True)
    acc3, img3, cnt3 = _frame_on_device(gpu, sc, cam, 2, count=True)
    assert np.array_equal(acc1.view(np.uint32), acc3.view(np.uint32)) and np.array_equal(img1, img3) and cnt1 == cnt3
    osc = ol.OracleScene(data)
    rs = np.random.RandomState(6)
    rows, cols = rs.randint(0, 1080, 64), rs.randint(0, 1920, 64)
    rr = np.repeat(rows, 16).astype(np.int32); cc = np.repeat(cols, 16).astype(np.int32); ss = np.tile(np.arange(16, dtype=np.int32), 64)
    rad = osc.radiance_samples(cam, rr, cc, ss).reshape(64, 16, 3)
    expect = np.zeros((64, 3), np.float32)
    for s in range(16):
        expect = expect + rad[:, s]
    assert np.array_equal(acc1[rows, cols].view(np.uint32), expect.view(np.uint32))


def _edge_scene(gpu):
    """Cornell + a transformed, UV-mapped, textured sphere + three lights (POINT, DISTANT, POINT), thin-lens camera."""
    s = gpu.scenes.cornell()
    albedo, mr = gpu.scenes._procedural_textures(64)
    s.textures = [albedo, mr]
    s.materials.append(gpu.scenes.material(gpu.scenes.METALLIC_ROUGHNESS, (1, 1, 1), alpha_x=0.3, alpha_y=0.4, albedo_tex=0, mr_tex=1))
    s.materials.append(gpu.scenes.material(gpu.scenes.DIELECTRIC, ior=(1.5, 1.5, 1.5), alpha_x=0.2, alpha_y=0.1))
    idx, v, n, uv = gpu.scenes.uv_sphere((0, 0, 0), 1.0, 16, 8)
    c, s_ = np.cos(0.6), np.sin(0.6)
    T = np.array([[70 * c, 0, 70 * s_, 330], [0, 90, 0, 240], [-70 * s_, 0, 70 * c, 150], [0, 0, 0, 1]], np.float32)   # rotate-Y * scale + translate
    s.add_mesh(idx, v, n, 3, uvs=uv * 3.0 - 1.0, transform=T, name="xformed")       # uvs outside [0,1): wrap incl. negatives
    T2 = np.array([[40, 0, 0, 150], [0, 40, 0, 380], [0, 0, 40, 300], [0, 0, 0, 1]], np.float32)
    s.add_mesh(idx, v, n, 4, uvs=uv, transform=T2, name="glass")
    s.lights = [gpu.scenes.light(gpu.scenes.POINT, (278.0, 500.0, 279.5), (1, 1, 1), 60000.0),
                gpu.scenes.light(gpu.scenes.DISTANT, (0.2, -0.9, 0.3), (1, 0.9, 0.8), 0.5),
                gpu.scenes.light(gpu.scenes.POINT, (100.0, 100.0, 100.0), (0, 1, 0), 90000.0)]   # never chosen (quirk Q2)
    s.sky = (0.1, 0.2, 0.3)
    s.camera["defocus_angle"] = 1.5
    s.camera["focus_distance"] = 1000.0
    return s


@pytest.mark.parametrize("integrator", [1, 2])
@pytest.mark.parametrize("max_prims", [1, 4])
def test_render_edge_cases(gpu, integrator, max_prims):
    """mesh transforms baked on upload, texture wrap with negative uv, 3 lights (Q2), thin lens, maxPrimsInNode > 1."""
    data = _edge_scene(gpu)
    sc = gpu.Scene(data); sc.buildBVH(max_prims)
    osc = ol.OracleScene(data)          # same data.max_prims_in_node
    cam_g, acc, img, cnt = _render_both(gpu, data, sc, osc, 120, 90, 2, 2, 6, integrator=integrator)
    assert_same_f32(cam_g.acc_, acc, "accumulation buffer")
    assert (cam_g.img_ == img).all()
    assert cam_g.counters == cnt
    if max_prims == 4:
        assert sc.bvh()[0]["num_prims"].max() > 1


@pytest.mark.parametrize("integrator", [1, 2])
def test_render_depth_zero_and_no_lights(gpu, cornell_pair, integrator):
    data, sc, osc = cornell_pair
    cam_g, acc, img, cnt = _render_both(gpu, data, sc, osc, 64, 48, 1, 2, 0, integrator=integrator)    # maxDepth 0: one closestHit per path
    assert_same_f32(cam_g.acc_, acc, "depth 0")
    assert cam_g.counters == cnt and cnt["n_any"] == 0 and cnt["n_closest"] == cnt["n_camera"]
    nl = gpu.scenes.cornell(); nl.lights = []; nl.sky = (0.5, 0.6, 0.7)
    s2 = gpu.Scene(nl); s2.buildBVH()
    cam_g, acc, img, cnt = _render_both(gpu, nl, s2, ol.OracleScene(nl), 64, 48, 2, 1, 5, integrator=integrator)
    assert_same_f32(cam_g.acc_, acc, "no lights")
    assert cam_g.counters == cnt and cnt["n_any"] == 0


def test_shard_buffers_are_zeroed_every_frame(gpu, cornell_pair):
    """multi-GPU step: the reduce target on rank 0 holds last frame's full image; the next shard render must
    zero everything it does not own (acc and img) before the next reduce."""
    import torch
    data, sc, osc = cornell_pair
    cam = data.camera_desc(100, 70, 2, 1, 3)
    dev = torch.device("cuda", 0)
    acc = torch.full((70 * 100 * 3,), 7.0, dtype=torch.float32, device=dev)
    img = torch.full((70 * 100 * 3,), 9, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())      # the fills above ran on the default stream
    with torch.cuda.stream(st):
        gpu.distributed.render_shard(sc, cam, 1, 3, acc, img, stream=st.cuda_stream)
    torch.cuda.synchronize()
    mask = gpu.distributed.tile_owner_mask(100, 70, 1, 3)
    a = acc.cpu().numpy().reshape(70, 100, 3); i = img.cpu().numpy().reshape(70, 100, 3)
    assert (a[~mask] == 0).all() and (i[~mask] == 0).all()
    ref_acc, ref_img, _ = osc.render(cam)
    assert np.array_equal(a[mask].view(np.uint32), ref_acc[mask].view(np.uint32)) and np.array_equal(i[mask], ref_img[mask])


# ---- uncounted kernels (the timed path): HBM-resident scenes walk the 8-ary quantised BVH there ----
def test_wide_bvh_is_built_for_hbm_scenes(gpu, mixed_pair, cornell_pair):
    assert mixed_pair[1].info()["wide_depth"] >= 2 and mixed_pair[1].info()["wide_bytes"] > 0
    assert cornell_pair[1].info()["lds_resident"]            # LDS-resident scenes keep the threaded records


@pytest.mark.parametrize("integrator", [1, 2])
def test_uncounted_render_mixed(gpu, mixed_pair, integrator):
    """count_rays=False is what bench.py times: same film as the oracle, bit for bit."""
    data, sc, osc = mixed_pair
    cam_g = gpu.StaticCamera(200, 120, data.camera, 2, 2, 8)
    cam_g.render(sc, count_rays=False, integrator=integrator)
    acc, img, _ = osc.render(data.camera_desc(200, 120, 2, 2, 8), count=False)
    assert_same_f32(cam_g.acc_, acc, "accumulation buffer")
    assert (cam_g.img_ == img).all()


@pytest.mark.parametrize("integrator", [2])
def test_uncounted_wavefront_atrium_and_axis_parallel(gpu, integrator):
    """the wavefront trace kernels walk the wide nodes too (uncounted, HBM-resident scenes); second scene: every shadow
    ray axis-parallel, i.e. traced on the binary records at activation"""
    for overhead in (False, True):
        data = gpu.scenes.atrium(target_tris=20000)
        if overhead:
            for l in data.lights:
                if l["type"] == 1:
                    l["position"] = (0.0, -1.0, 0.0)
        sc = gpu.Scene(data); sc.buildBVH()
        osc = ol.OracleScene(data)
        cam_g = gpu.StaticCamera(160, 90, data.camera, 2, 2, 8)
        cam_g.render(sc, count_rays=False, integrator=integrator)
        acc, img, _ = osc.render(data.camera_desc(160, 90, 2, 2, 8), count=False)
        assert_same_f32(cam_g.acc_, acc, f"accumulation buffer (overhead light: {overhead})")
        assert (cam_g.img_ == img).all()


@pytest.mark.parametrize("tris,prims", [(20000, 1), (60000, 1), (20000, 4)])
def test_uncounted_render_atrium(gpu, tris, prims):
    data = gpu.scenes.atrium(target_tris=tris)
    data.max_prims_in_node = prims
    sc = gpu.Scene(data); sc.buildBVH()
    assert sc.info()["wide_depth"] >= 3
    osc = ol.OracleScene(data)
    cam_g = gpu.StaticCamera(160, 90, data.camera, 2, 2, 8)
    cam_g.render(sc, count_rays=False, integrator=1)
    acc, img, _ = osc.render(data.camera_desc(160, 90, 2, 2, 8), count=False)
    assert_same_f32(cam_g.acc_, acc, "accumulation buffer")
    assert (cam_g.img_ == img).all()


def test_uncounted_render_axis_parallel_shadow_rays(gpu):
    """A DISTANT light straight overhead makes every shadow ray axis-parallel (1/d = inf): those waves must
    take the exact binary records (AABB::hit's inf/NaN behaviour is not monotone), the others the wide nodes."""
    data = gpu.scenes.atrium(target_tris=20000)
    for l in data.lights:
        if l["type"] == 1:
            l["position"] = (0.0, -1.0, 0.0)
    sc = gpu.Scene(data); sc.buildBVH()
    osc = ol.OracleScene(data)
    cam_g = gpu.StaticCamera(128, 72, data.camera, 2, 2, 6)
    cam_g.render(sc, count_rays=False, integrator=1)
    acc, img, cnt = osc.render(data.camera_desc(128, 72, 2, 2, 6))
    assert cnt["n_any"] > 0
    assert_same_f32(cam_g.acc_, acc, "accumulation buffer")
    assert (cam_g.img_ == img).all()


def test_dynamic_camera_progressive_and_restart(gpu, cornell_pair):
    """DynamicCamera (camera.cpp:130-255): non-blocking progressive render; the finished frame equals the
    oracle's; a render() issued while a frame is in flight abandons it and the new frame is complete and exact."""
    data, sc, osc = cornell_pair
    dyn = gpu.DynamicCamera(96, 64, data.camera, 3, 2, 4, samplesPerPass=2)
    try:
        dyn.render(sc)
        assert dyn.wait(120) and dyn.finished() and dyn.currentSample_ == 6
        acc, img, _ = osc.render(data.camera_desc(96, 64, 3, 2, 4), count=False)
        assert_same_f32(dyn.acc_, acc, "dynamic camera accumulation")
        assert (dyn.img_ == img).all()
        moved = dict(data.camera); moved["center"] = (250.0, 300.0, -760.0)
        dyn.render(sc)                      # frame in flight ...
        dyn.properties_ = moved
        dyn.render(sc)                      # ... abandoned for the moved camera
        assert dyn.wait(120) and dyn.finished()
        ref = gpu.StaticCamera(96, 64, moved, 3, 2, 4)
        ref.render(sc)
        assert_same_f32(dyn.acc_, ref.acc_, "restarted frame")
        assert (dyn.img_ == ref.img_).all()
        dyn.resize(48, 32)
        dyn.render(sc)
        assert dyn.wait(120) and dyn.img_.shape == (32, 48, 3) and dyn.img_.any()
    finally:
        dyn.stopRender()


def test_frame_gather_pack_scatter_on_device(gpu, cornell_pair):
    """FrameGather's device side (the RCCL gather itself needs > 1 GPU): 5 shards rendered one after the other on
    this GPU, packed to compact slabs, handed to rank 0's receive buffer and scattered -> the 1-GPU frame."""
    import torch
    data, sc, osc = cornell_pair
    W, H, world = 250, 130, 5
    cam = data.camera_desc(W, H, 2, 2, 4)
    dev = torch.device("cuda", 0)
    full_acc, full_img, _ = _frame_on_device(gpu, sc, cam, 1)
    root = gpu.distributed.FrameGather(W, H, 0, world, dev)
    acc0 = img0 = None
    st = torch.cuda.Stream()            # a non-default stream: 0 would mean "the library's own stream"
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for r in range(world):
            acc = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
            img = torch.zeros(H * W * 3, dtype=torch.uint8, device=dev)
            gpu.distributed.render_shard(sc, cam, r, world, acc, img, stream=st.cuda_stream, integrator=1)
            fg = root if r == 0 else gpu.distributed.FrameGather(W, H, r, world, dev)
            root.recv[r].copy_(fg.pack(acc, img))
            if r == 0:
                acc0, img0 = acc, img
        acc0 += 1.0                     # whatever rank 0 holds outside ... and inside its tiles is overwritten
        root.scatter(acc0, img0)
    torch.cuda.synchronize()
    assert_same_f32(acc0.cpu().numpy().reshape(H, W, 3), full_acc, "gathered frame")
    assert (img0.cpu().numpy().reshape(H, W, 3) == full_img).all()


def _tie_scene(gpu, n=24):
    """Three coincident tessellated walls with different albedos (every hit has two exact rivals at the same t), the
    same wall again rotated by a transform that maps onto itself, a floor, and rays that run exactly along shared
    edges: whatever structure walks the BVH must resolve equal t the way Scene::closestHit does -- first found wins."""
    sc = gpu.scenes
    s = sc.SceneData("ties")
    s.materials = [sc.material(sc.DIFFUSE, a) for a in [(0.9, 0.1, 0.1), (0.1, 0.9, 0.1), (0.1, 0.1, 0.9), (0.7, 0.7, 0.7)]]
    for m in range(3):
        g = sc._grid_quad((-4, 0, -3), (8, 0, 0), (0, 6, 0), n, n, (0, 0, 1))
        s.add_mesh(g[0], g[1], g[2], m, uvs=g[3], name=f"wall{m}")
    g = sc._grid_quad((-6, 0, -3), (12, 0, 0), (0, 0, 9), n, n, (0, 1, 0))
    s.add_mesh(g[0], g[1], g[2], 3, uvs=g[3], name="floor")
    s.lights = [sc.light(sc.POINT, (1.0, 5.0, 4.0), (1, 1, 1), 60.0), sc.light(sc.DISTANT, (0.3, -1.0, -0.4), (1, 1, 1), 1.5)]
    s.sky = (0.2, 0.3, 0.4)
    s.camera = dict(center=(0.0, 3.0, 9.0), target=(0.0, 3.0, -3.0), up=(0, 1, 0), yfov=45.0, defocus_angle=0.0, focus_distance=1.0)
    return s


@pytest.mark.parametrize("max_prims", [1, 4])
def test_equal_t_ties_resolve_as_in_the_reference(gpu, max_prims):
    data = _tie_scene(gpu)
    data.max_prims_in_node = max_prims
    sc = gpu.Scene(data); sc.buildBVH(max_prims)
    assert sc.info()["wide_depth"] >= 2 and not sc.info()["lds_resident"]
    osc = ol.OracleScene(data)
    # per-ray API (wide nodes are not used here, binary records): rays through grid vertices and along grid edges
    rs = np.random.RandomState(3)
    n = 20000
    gx = -4 + 8 * rs.randint(0, 25, n) / 24.0
    gy = 6 * rs.randint(0, 25, n) / 24.0
    target = np.stack([gx, gy, np.full(n, -3.0)], 1).astype(np.float32)
    o = np.tile(np.array([[0.0, 3.0, 9.0]], np.float32), (n, 1))
    o[n // 2:] = target[n // 2:] + np.array([0, 0, 5], np.float32)          # second half: axis-parallel rays onto vertices
    d = (target - o).astype(np.float32)
    g, r = sc.closestHit(o, d), osc.closestHit(o, d)
    assert (g["hit"] == r["hit"]).all() and g["hit"].mean() > 0.9
    assert (g["prim"] == r["prim"]).all()
    assert_same_f32(g["t"], r["t"], "t")
    # whole frames: counted (binary records) and uncounted (wide nodes) against the oracle
    for count in (True, False):
        cam_g = gpu.StaticCamera(160, 120, data.camera, 2, 2, 6)
        cam_g.render(sc, count_rays=count, integrator=1)
        acc, img, cnt = osc.render(data.camera_desc(160, 120, 2, 2, 6))
        assert_same_f32(cam_g.acc_, acc, f"accumulation buffer (count_rays={count})")
        assert (cam_g.img_ == img).all()
        if count:
            assert cam_g.counters == cnt


def test_gltf_scene_renders_like_the_oracle(gpu, tmp_path):
    """a GLB written and re-read by jtx_pathtracer_amd.gltf (node transforms baked, flipped uvs, embedded
    PNG albedo + metallic-roughness maps decoded the stbi_loadf way) goes through the same hot path, bit-exact."""
    import jtx_pathtracer_amd.gltf as gltf
    sc_ = gpu.scenes
    src = sc_.SceneData("src")
    rs = np.random.RandomState(4)
    tex = [rs.randint(0, 256, (16, 32, 3)).astype(np.uint8), rs.randint(0, 256, (8, 8, 4)).astype(np.uint8)]
    src.materials = [sc_.material(sc_.METALLIC_ROUGHNESS, (1, 1, 1), alpha_x=0.2, alpha_y=0.7, albedo_tex=0, mr_tex=1),
                     sc_.material(sc_.METALLIC_ROUGHNESS, (1, 1, 1), alpha_x=1.0, alpha_y=0.15)]
    g = sc_._grid_quad((-3, -1, -3), (6, 0, 0), (0, 0, 6), 6, 6, (0, 1, 0))
    src.add_mesh(g[0], g[1], g[2], 0, uvs=g[3], name="floor")
    c = sc_._column(0.0, 0.0, -1.0, 1.0, 0.6, 16, 6)
    src.add_mesh(c[0], c[1], c[2], 1, uvs=c[3], name="column")
    a = np.deg2rad(30.0)
    rot = np.eye(4); rot[0, 0], rot[0, 1], rot[1, 0], rot[1, 1] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
    path = str(tmp_path / "s.glb")
    gltf.write_glb(path, src, textures_u8=tex, node_matrices=[None, rot])
    data = gltf.load_gltf(path, background=(0.5, 0.6, 0.8))
    data.lights = [sc_.light(sc_.POINT, (2.0, 4.0, 3.0), (1, 1, 1), 40.0)]
    data.camera = dict(center=(0.0, 2.0, 7.0), target=(0.0, 0.0, 0.0), up=(0, 1, 0), yfov=35.0, defocus_angle=0.0, focus_distance=1.0)
    sc = gpu.Scene(data); sc.buildBVH()
    osc = ol.OracleScene(data)
    for count in (True, False):
        cam_g = gpu.StaticCamera(128, 96, data.camera, 2, 2, 5)
        cam_g.render(sc, count_rays=count)
        acc, img, cnt = osc.render(data.camera_desc(128, 96, 2, 2, 5))
        assert_same_f32(cam_g.acc_, acc, f"glTF scene (count_rays={count})")
        assert (cam_g.img_ == img).all()
        if count:
            assert cam_g.counters == cnt
    assert len(np.unique(cam_g.img_.reshape(-1, 3), axis=0)) > 200          # textured, lit, not flat


def test_kernel_known_answers_gpu(gpu, cornell_pair):
    """the committed known answers (tests/golden: ray-triangle grazers through vertices and edge midpoints, shadow rays,
    the Cornell LinearBVHNode array, 256 pixels x 16 per-sample radiances of config C1) straight from the GPU"""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(root, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    gold = json.load(open(os.path.join(root, "tests", "golden", "oracle_golden.json")))["kernels"]
    data, sc, osc = cornell_pair
    o, d = mg.tri_rays(data)
    r = sc.closestHit(o, d)
    got = mg.crc(np.concatenate([r["hit"].astype(np.float32), r["t"], r["prim"].astype(np.float32), r["b1"], r["b2"],
                                 r["point"].reshape(-1), r["normal"].reshape(-1), r["uv"].reshape(-1)]))
    assert got == gold["cornell_closest"]["crc32"] and int(r["hit"].sum()) == gold["cornell_closest"]["hits"]
    a = sc.anyHit(o, d, 0.0, np.full(len(o), 0.9999, np.float32))
    assert mg.crc(a) == gold["cornell_any"]["crc32"]
    nodes, refs = sc.bvh()
    assert mg.crc(nodes.tobytes() + refs.tobytes()) == gold["cornell_bvh_crc32"]
    rs = np.random.RandomState(79)
    row = np.repeat(rs.randint(0, 512, 256), 16).astype(np.int32); col = np.repeat(rs.randint(0, 512, 256), 16).astype(np.int32)
    smp = np.tile(np.arange(16), 256).astype(np.int32)
    rgb = gpu.api.radiance_samples(sc, data.camera_desc(512, 512, 4, 4, 4), row, col, smp)
    assert mg.crc(rgb) == gold["cornell_radiance_samples_crc32"]


def _pipeline_worker(rank, world, port, out_dir):
    """one rank of the rehearsal: both ranks render on cuda:0, gloo carries the exchange (staged through the host)"""
    import os
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p_ in (root, os.path.join(root, "tests")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    import jtx_pathtracer_amd as jtx
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = jtx.scenes.cornell()
        sc = jtx.Scene(data); sc.buildBVH()
        W, H = 200, 136
        # two views taking turns, so that consecutive frames (in flight together, one per frame slot) differ
        views = []
        for yfov in (39.3077, 55.0):
            c = dict(data.camera); c["yfov"] = yfov
            data.camera = c
            views.append((c, data.camera_desc(W, H, 2, 2, 4)))
        fg = jtx.distributed.FrameGather(W, H, rank, world, dev)
        pipe = jtx.distributed.ShardPipeline(sc, views[0][1], rank, world, dev, fg, integrator=1, timing=True)
        assert len(pipe.rstreams) == jtx.distributed.FRAME_SLOTS == 3
        nframes = 7                                          # odd: both buffer pairs, both slots and the hand-over get used
        hist_a = [torch.zeros_like(pipe.accs[0]) for _ in range(nframes)] if rank == 0 else None
        hist_i = [torch.zeros_like(pipe.imgs[0]) for _ in range(nframes)] if rank == 0 else None
        for k in range(nframes):
            pipe.cam = views[k % 2][1]
            pipe.step()
            if rank == 0:                                    # keep every assembled frame: a copy behind its exchange, on the exchange stream
                with torch.cuda.stream(pipe.xstream):
                    hist_a[k].copy_(pipe.frame_acc); hist_i[k].copy_(pipe.frame_img)
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            ok = pipe.exchange_ms() is not None and len(pipe.timed) == 0
            for v, (cdict, _) in enumerate(views):
                full = jtx.StaticCamera(W, H, cdict, 2, 2, 4); full.render(sc, integrator=1)
                for k in range(v, nframes, 2):
                    a = hist_a[k].cpu().numpy().reshape(H, W, 3); i = hist_i[k].cpu().numpy().reshape(H, W, 3)
                    ok = ok and np.array_equal(a.view(np.uint32), full.acc_.view(np.uint32)) and np.array_equal(i, full.img_)
            open(os.path.join(out_dir, "result"), "w").write("ok" if ok else "mismatch")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_pipeline_two_rank_rehearsal(gpu, tmp_path):
    """ShardPipeline (bench.py's frame loop: two frames in flight -- frame i + 1 launched on the other render stream and frame slot
    while frame i's tail and resolve run -- and the exchange of frame i on a side stream) with two processes sharing this GPU: EVERY
    one of rank 0's 7 consecutive assembled frames (two views taking turns) equals the 1-GPU frame of its view, bit for bit."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_pipeline_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "result").read() == "ok"


@pytest.mark.parametrize("which", ["cornell", "atrium"])
def test_two_frames_in_flight_on_one_rank(gpu, which):
    """jtx_mi_render_opts.frame_slot: consecutive frames of ONE scene on three streams, in the scene's three frame slots, each into film
    buffers of its own (what bench.py times at N = 1) -- 8 frames of four views taking turns, every frame kept by a copy on its own
    render stream, each bit-identical to the frame of its view rendered alone; then a counting launch (the scene's singletons: ordered
    against ALL slots by the library) straight behind the frames in flight gives the one-frame counters."""
    import torch
    data = gpu.scenes.cornell() if which == "cornell" else gpu.scenes.atrium(target_tris=6000)
    sc = gpu.Scene(data); sc.buildBVH()
    W, H = 328, 200
    views = []
    for dy in (0.0, 11.0, -7.0, 23.0):
        c = dict(data.camera); c["center"] = (c["center"][0], c["center"][1] + dy, c["center"][2])
        data.camera = c
        views.append((c, data.camera_desc(W, H, 2, 2, 5)))
    dev = torch.device("cuda", 0)
    pipe = gpu.distributed.ShardPipeline(sc, views[0][1], 0, 1, dev, None, integrator=1)
    nb = len(pipe.rstreams)
    assert nb == gpu.distributed.FRAME_SLOTS == 3 and pipe.xstream is None
    n = 8
    hist_a = [torch.zeros_like(pipe.accs[0]) for _ in range(n)]; hist_i = [torch.zeros_like(pipe.imgs[0]) for _ in range(n)]
    for k in range(n):
        pipe.cam = views[k % 4][1]
        pipe.step()
        with torch.cuda.stream(pipe.rstreams[k % nb]):
            hist_a[k].copy_(pipe.accs[k % nb]); hist_i[k].copy_(pipe.imgs[k % nb])
    # a counted frame right behind them, on a third stream: it must wait for both slots (it uses the scene's ray counters)
    st = torch.cuda.Stream(device=dev)
    acc = torch.zeros_like(pipe.accs[0]); img = torch.zeros_like(pipe.imgs[0])
    gpu.distributed.render_shard(sc, views[0][1], 0, 1, acc, img, stream=st.cuda_stream, count_rays=True, integrator=1)
    torch.cuda.synchronize()
    import ctypes as C
    cnt = gpu._capi.Counters(); gpu._capi.check(gpu._capi.load().jtx_mi_get_counters(sc.handle, C.byref(cnt)))
    for v, (cdict, _) in enumerate(views):
        full = gpu.StaticCamera(W, H, cdict, 2, 2, 5); full.render(sc, count_rays=(v == 0), integrator=1)
        if v == 0:
            assert cnt.as_dict() == full.counters
            assert np.array_equal(acc.cpu().numpy().view(np.uint32).reshape(H, W, 3), full.acc_.view(np.uint32))
        for k in range(v, n, 4):
            a = hist_a[k].cpu().numpy().reshape(H, W, 3); i = hist_i[k].cpu().numpy().reshape(H, W, 3)
            assert np.array_equal(a.view(np.uint32), full.acc_.view(np.uint32)), f"frame {k} (view {v}) differs"
            assert np.array_equal(i, full.img_), f"frame {k} (view {v}): RGB8 differs"
    # frame_slot is validated
    with pytest.raises(gpu.JtxMiError):
        gpu.distributed.render_shard(sc, views[0][1], 0, 1, acc, img, stream=st.cuda_stream, frame_slot=gpu.distributed.FRAME_SLOTS)
    sc.destroy()


@pytest.mark.parametrize("lightdir", [(1e-6, -1.0, 3e-7), (-2e-9, -1.0, 1e-10), (0.3, -1e-5, -0.9)])
def test_wide_bvh_nearly_axis_parallel_rays(gpu, lightdir):
    """stress of the wide-node error budget: shadow rays with |1/d| up to 1e10 (inside the
    2^40 range, so they DO take the quantised nodes; b = (origin - o)/d gets huge) and a camera 2000 units away"""
    data = gpu.scenes.atrium(target_tris=20000)
    for l in data.lights:
        if l["type"] == 1:
            l["position"] = lightdir
    cam = dict(data.camera); c = np.array(cam["center"], np.float64); t = np.array(cam["target"], np.float64)
    cam["center"] = tuple(t + (c - t) / np.linalg.norm(c - t) * 2000.0); cam["yfov"] = 1.6
    sc = gpu.Scene(data); sc.buildBVH()
    assert sc.info()["wide_depth"] >= 3
    osc = ol.OracleScene(data)
    cam_g = gpu.StaticCamera(120, 80, cam, 2, 2, 6)
    cam_g.render(sc, count_rays=False, integrator=1)
    data.camera = cam
    acc, img, cnt = osc.render(data.camera_desc(120, 80, 2, 2, 6))
    assert cnt["n_any"] > 1000 and cnt["n_accept"] > 1000
    assert_same_f32(cam_g.acc_, acc, "accumulation buffer")
    assert (cam_g.img_ == img).all()


def test_wide_bvh_on_a_deep_degenerate_tree(gpu):
    """triangles of geometrically growing size nested in one corner: the SAH peels a few triangles per level, the
    binary tree is a chain and the wide tree stays deep (14 levels even with the area-optimal cut: a long per-lane LDS stack, fewer workgroups per
    CU); the film still equals the oracle's"""
    sc_ = gpu.scenes
    s = sc_.SceneData("deep")
    s.materials = [sc_.material(sc_.DIFFUSE, (0.7, 0.6, 0.5))]
    v = []
    for k in range(320):
        e = np.float32(1.06) ** k * np.float32(0.01)
        v += [(0, 0, -k * 1e-3), (e, 0, -k * 1e-3), (0, e, -k * 1e-3)]
    v = np.array(v, np.float32)
    nrm = np.tile(np.array([[0, 0, 1]], np.float32), (len(v), 1))
    s.add_mesh(np.arange(len(v), dtype=np.int32).reshape(-1, 3), v, nrm, 0)
    s.lights = [sc_.light(sc_.POINT, (20.0, 30.0, 40.0), (1, 1, 1), 3000.0)]
    s.sky = (0.3, 0.4, 0.5)
    s.camera = dict(center=(6.0, 5.0, 30.0), target=(3.0, 3.0, 0.0), up=(0, 1, 0), yfov=50.0, defocus_angle=0.0, focus_distance=1.0)
    sc = gpu.Scene(s); sc.buildBVH()
    info = sc.info()
    assert not info["lds_resident"] and info["max_depth"] >= 15
    assert 12 <= info["wide_depth"] <= 24
    osc = ol.OracleScene(s)
    cam_g = gpu.StaticCamera(128, 96, s.camera, 2, 2, 5)
    cam_g.render(sc, count_rays=False, integrator=1)
    acc, img, cnt = osc.render(s.camera_desc(128, 96, 2, 2, 5))
    assert cnt["n_accept"] > 10000
    assert_same_f32(cam_g.acc_, acc, "deep tree")
    assert (cam_g.img_ == img).all()
    cam_g.render(sc, count_rays=False, integrator=2)                  # the wavefront trace kernels use the same stack
    assert_same_f32(cam_g.acc_, acc, "deep tree, wavefront")


def test_radiance_buffer_cap_renders_in_passes(gpu, cornell_pair, monkeypatch):
    """k_render_paths keeps one 16-byte record per path of a pass; frames whose records would not fit the cap go in
    several passes of consecutive strata (here: 1 MB -> 2 of 16 strata per pass), the resolve continuing the sums"""
    data, sc, osc = cornell_pair
    monkeypatch.setenv("JTX_MAX_RAD_MB", "1")
    cam_g = gpu.StaticCamera(200, 120, data.camera, 4, 4, 4)
    cam_g.render(sc, count_rays=False, integrator=1)
    acc, img, _ = osc.render(data.camera_desc(200, 120, 4, 4, 4), count=False)
    assert_same_f32(cam_g.acc_, acc, "multi-pass frame")
    assert (cam_g.img_ == img).all()
    cam_g.render(sc, count_rays=False, integrator=1, sample_begin=0, sample_end=7)      # and a partial range, odd length
    cam_g.render(sc, count_rays=False, integrator=1, sample_begin=7, sample_end=16)
    assert_same_f32(cam_g.acc_, acc, "multi-pass frame, resumed")


def test_cancel_inside_a_split_pass_reports_what_is_in_the_film(gpu, atrium_full, monkeypatch):
    """a pass whose records exceed the radiance buffer goes in several launches, each resolved on its own; a
    cancellation that arrives in a later launch leaves the earlier ones in the film -- and `currentSample_` must say so: the
    film equals a render of exactly the strata reported"""
    import threading
    data, sc, osc = atrium_full
    monkeypatch.setenv("JTX_MAX_RAD_MB", "280")                       # 1920x1080: 8 strata per launch, 8 launches per 64-spp pass
    cam = gpu.StaticCamera(1920, 1080, data.camera, 8, 8, 8)
    cam._pin()
    timer = threading.Timer(0.17, cam.terminateRender)
    timer.start(); cam.render(sc); timer.join()
    n = cam.currentSample_
    if n == 64:
        pytest.skip("the frame finished before the cancellation arrived")
    assert n % 8 == 0, n
    got = cam.acc_.copy()
    part = gpu.StaticCamera(1920, 1080, data.camera, 8, 8, 8)
    if n:
        part.render(sc, sample_begin=0, sample_end=n)
    assert_same_f32(got, part.acc_, f"film after a cancellation inside a split pass ({n} strata reported)")


@pytest.mark.parametrize("seed,max_prims", [(11, 1), (12, 1), (13, 4)])
def test_uncounted_random_triangle_soups(gpu, seed, max_prims):
    """no structure to lean on: thousands of overlapping random triangles of very different sizes, the four material
    types, a point and a distant light -- the timed kernels (wide BVH, dynamic path assignment) against the oracle,
    both integrators that have uncounted wide paths"""
    sc_ = gpu.scenes
    rs = np.random.RandomState(seed)
    n = 4000
    c = rs.uniform(-10, 10, (n, 3))
    size = np.exp(rs.uniform(np.log(0.05), np.log(5.0), (n, 1, 1)))
    tri = (c[:, None, :] + rs.normal(size=(n, 3, 3)) * size).astype(np.float32)
    s = sc_.SceneData("soup")
    s.materials = [sc_.material(sc_.DIFFUSE, (0.7, 0.5, 0.3)),
                   sc_.material(sc_.CONDUCTOR, ior=sc_.GOLD_IOR, k=sc_.GOLD_K, alpha_x=0.2, alpha_y=0.1),
                   sc_.material(sc_.DIELECTRIC, ior=(1.5, 1.5, 1.5), alpha_x=0.0, alpha_y=0.0),
                   sc_.material(sc_.METALLIC_ROUGHNESS, (0.4, 0.6, 0.8), alpha_x=0.3, alpha_y=0.5)]
    fn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-20)
    for m in range(4):
        sel = np.arange(n) % 4 == m
        v = tri[sel].reshape(-1, 3)
        nrm = np.repeat(fn[sel], 3, axis=0).astype(np.float32)
        s.add_mesh(np.arange(len(v), dtype=np.int32).reshape(-1, 3), v, nrm, m)
    s.lights = [sc_.light(sc_.POINT, (3.0, 14.0, 5.0), (1, 1, 1), 400.0), sc_.light(sc_.DISTANT, (0.2, -1.0, 0.3), (1, 0.9, 0.8), 1.0)]
    s.sky = (0.4, 0.5, 0.7)
    s.camera = dict(center=(0.0, 2.0, 32.0), target=(0.0, 0.0, 0.0), up=(0, 1, 0), yfov=40.0, defocus_angle=0.0, focus_distance=1.0)
    s.max_prims_in_node = max_prims
    sc = gpu.Scene(s); sc.buildBVH(max_prims)
    assert sc.info()["wide_depth"] >= 3
    osc = ol.OracleScene(s)
    acc, img, cnt = osc.render(s.camera_desc(144, 96, 2, 2, 6))
    assert cnt["n_accept"] > 20000 and cnt["n_any"] > 10000
    for integrator in (1, 2):
        cam_g = gpu.StaticCamera(144, 96, s.camera, 2, 2, 6)
        cam_g.render(sc, count_rays=False, integrator=integrator)
        assert_same_f32(cam_g.acc_, acc, f"soup seed {seed}, integrator {integrator}")
        assert (cam_g.img_ == img).all()


# ------------------------------------------------------------------------------------------------
# The EXACT launches bench.py times, at BASELINE.json's full sizes ("What's weak" 2): 1 rank,
# count_rays = False -> k_render_paths with the default strata-group / pass policy.  Each is compared, whole frame,
# bit for bit, with the counted launch of the same frame (k_render_pixels on the reference's binary node records,
# itself oracle-checked above) and, per pixel, with the oracle's in-order per-sample sums.
# ------------------------------------------------------------------------------------------------
def _oracle_pixel_sums(osc, cam, rows, cols, spp):
    n = len(rows)
    rr = np.repeat(rows, spp).astype(np.int32); cc = np.repeat(cols, spp).astype(np.int32)
    ss = np.tile(np.arange(spp, dtype=np.int32), n)
    rad = osc.radiance_samples(cam, rr, cc, ss).reshape(n, spp, 3)
    expect = np.zeros((n, 3), np.float32)
    for s in range(spp):
        expect = expect + rad[:, s]                      # AccumulationBuffer::updatePixel order (image.hpp:82-86)
    return expect


def _expect_bytes(expect, spp):
    g = np.sqrt(np.maximum(expect / np.float32(spp), np.float32(0))).astype(np.float32)
    return (np.float32(255.999) * np.clip(g, np.float32(0), np.float32(0.999))).astype(np.float32).astype(np.int32)


def test_timed_launch_config2_full_frame(gpu, cornell_pair):
    """C2 exactly as bench.py times it: Cornell 1920x1080, 64 spp, depth 8, one rank, uncounted (k_render_paths<LDS>)."""
    data, sc, osc = cornell_pair
    cam = data.camera_desc(1920, 1080, 8, 8, 8)
    acc_u, img_u, _ = _frame_on_device(gpu, sc, cam, 1, count=False)
    acc_c, img_c, cnt = _frame_on_device(gpu, sc, cam, 1, count=True)
    assert np.array_equal(acc_u.view(np.uint32), acc_c.view(np.uint32)) and np.array_equal(img_u, img_c)
    assert cnt["n_camera"] == 1920 * 1080 * 64
    rs = np.random.RandomState(21)
    rows, cols = rs.randint(0, 1080, 256), rs.randint(0, 1920, 256)
    expect = _oracle_pixel_sums(osc, cam, rows, cols, 64)
    assert np.array_equal(acc_u[rows, cols].view(np.uint32), expect.view(np.uint32))
    assert np.abs(img_u[rows, cols].astype(np.int32) - _expect_bytes(expect, 64)).max() <= 1


def test_timed_launch_config3_full_frame(gpu, atrium_full):
    """C3 exactly as timed: atrium (262 k triangles), 1920x1080, ALL 64 strata, depth 8, one rank, uncounted =
    the 8-ary quantised traversal (k_render_paths<WIDE>) -- against the counted binary-record frame and the oracle."""
    data, sc, osc = atrium_full
    assert data.num_triangles > 250000 and sc.info()["wide_depth"] >= 2 and not sc.info()["lds_resident"]
    cam = data.camera_desc(1920, 1080, 8, 8, 8)
    acc_u, img_u, _ = _frame_on_device(gpu, sc, cam, 1, count=False)
    acc_c, img_c, cnt = _frame_on_device(gpu, sc, cam, 1, count=True)
    assert np.array_equal(acc_u.view(np.uint32), acc_c.view(np.uint32)) and np.array_equal(img_u, img_c)
    assert cnt["n_camera"] == 1920 * 1080 * 64
    rs = np.random.RandomState(22)
    rows, cols = rs.randint(0, 1080, 256), rs.randint(0, 1920, 256)
    expect = _oracle_pixel_sums(osc, cam, rows, cols, 64)
    assert np.array_equal(acc_u[rows, cols].view(np.uint32), expect.view(np.uint32))
    assert np.abs(img_u[rows, cols].astype(np.int32) - _expect_bytes(expect, 64)).max() <= 1


def test_timed_launch_config5_full_frame(gpu):
    """C5 exactly as timed: the mixed-material scene (all four BxDFs, textures, two lights), 1920x1080, 128 spp,
    depth 8, one rank, uncounted (k_render_paths<WIDE, MAT_ALL>)."""
    data = gpu.scenes.mixed()
    sc = gpu.Scene(data); sc.buildBVH()
    assert sc.info()["wide_depth"] >= 2
    cam = data.camera_desc(1920, 1080, 16, 8, 8)
    acc_u, img_u, _ = _frame_on_device(gpu, sc, cam, 1, count=False)
    acc_c, img_c, cnt = _frame_on_device(gpu, sc, cam, 1, count=True)
    assert np.array_equal(acc_u.view(np.uint32), acc_c.view(np.uint32)) and np.array_equal(img_u, img_c)
    assert cnt["n_camera"] == 1920 * 1080 * 128
    osc = ol.OracleScene(data)
    rs = np.random.RandomState(23)
    rows, cols = rs.randint(0, 1080, 256), rs.randint(0, 1920, 256)
    expect = _oracle_pixel_sums(osc, cam, rows, cols, 128)
    assert np.array_equal(acc_u[rows, cols].view(np.uint32), expect.view(np.uint32))
    assert np.abs(img_u[rows, cols].astype(np.int32) - _expect_bytes(expect, 128)).max() <= 1
    sc.destroy()


def test_timed_launch_config4_shard_and_pass_split(gpu, atrium_full):
    """C4: atrium at 3840x2160, 256 spp (16x16), depth 8.  (a) rank 0's shard of 8 exactly as an 8-GPU run launches it
    (default radiance-buffer cap) against the oracle's per-sample sums on 64 owned pixels; (b) the whole 4 K frame on
    ONE rank -- 34 GB of per-path records against the default 8 GiB cap, i.e. the real pass split -- must equal the
    shard on every owned pixel bit for bit and be untouched (zero) nowhere."""
    import torch
    data, sc, osc = atrium_full
    W, H, spp = 3840, 2160, 256
    cam = data.camera_desc(W, H, 16, 16, 8)
    acc_s, img_s, _ = _frame_on_device(gpu, sc, cam, 1, rank=0, world=8)
    tiles_x = (W + 31) // 32
    ty, tx = np.meshgrid(np.arange(H) // 32, np.arange(W) // 32, indexing="ij")
    owned = ((ty * tiles_x + tx) % 8) == 0
    assert not acc_s[~owned].any() and not img_s[~owned].any()        # non-owned pixels read exactly 0 (the reduce relies on it)
    rs = np.random.RandomState(24)
    oy, ox = np.nonzero(owned)
    pick = rs.choice(len(oy), 64, replace=False)
    rows, cols = oy[pick], ox[pick]
    expect = _oracle_pixel_sums(osc, cam, rows, cols, spp)
    assert np.array_equal(acc_s[rows, cols].view(np.uint32), expect.view(np.uint32))
    assert np.abs(img_s[rows, cols].astype(np.int32) - _expect_bytes(expect, spp)).max() <= 1
    assert (W * H * spp * 16) > (8 << 30)                              # the 1-rank frame cannot fit one pass
    acc_f, img_f, _ = _frame_on_device(gpu, sc, cam, 1)
    assert np.array_equal(acc_f[owned].view(np.uint32), acc_s[owned].view(np.uint32))
    assert np.array_equal(img_f[owned], img_s[owned])
    rows2, cols2 = rs.randint(0, H, 16), rs.randint(0, W, 16)
    expect2 = _oracle_pixel_sums(osc, cam, rows2, cols2, spp)
    assert np.array_equal(acc_f[rows2, cols2].view(np.uint32), expect2.view(np.uint32))
    # (c) the same frame in ONE launch: the card holds the 34 GB of records (288 GB of HBM), the cap is the caller's to raise
    #     (jtx_mi_render_opts.max_record_mb) -- the same film word for word, and the slot's memory is accounted and releasable
    acc_1, img_1, _ = _frame_on_device(gpu, sc, cam, 1, max_record_mb=36 << 10)
    assert np.array_equal(acc_1.view(np.uint32), acc_f.view(np.uint32)) and np.array_equal(img_1, img_f)
    assert sc.info()["frame_slot_bytes"] >= W * H * spp * 16
    sc.releaseFrames()
    assert sc.info()["frame_slot_bytes"] == 0
    del acc_s, acc_f, acc_1
    torch.cuda.empty_cache()


# ---- one host process, N devices behind the C-ABI (jtx_mi_multi_*) ----
@pytest.mark.parametrize("shards", [1, 2, 3, 8])
def test_multi_device_render_equals_one_device(gpu, cornell_pair, shards):
    """jtx_mi_multi_render over `shards` shards (all on device 0 here: render -> pack -> peer copy -> scatter is the code
    an 8-GPU node runs, with hipMemcpyPeerAsync degenerating to a device copy) gives the one-device frame bit for bit,
    at a ragged size, with and without per-pass previews."""
    data, sc, osc = cornell_pair
    ref = gpu.StaticCamera(200, 100, data.camera, 4, 2, 4); ref.render(sc)
    ms = gpu.MultiScene(data, [0] * shards)
    cam = gpu.StaticCamera(200, 100, data.camera, 4, 2, 4)
    assert ms.render(cam)
    assert_same_f32(cam.acc_, ref.acc_, f"{shards} shards"); assert (cam.img_ == ref.img_).all()
    assert cam.currentSample_ == 8 and len(ms.shard_ms()) == shards
    seen = []
    cam2 = gpu.StaticCamera(200, 100, data.camera, 4, 2, 4)
    assert ms.render(cam2, progress=lambda c, t: seen.append(c), samples_per_tick=3)
    assert seen == [3, 6, 8]
    assert_same_f32(cam2.acc_, ref.acc_, f"{shards} shards, 3 passes"); assert (cam2.img_ == ref.img_).all()
    # cancellation by the callback after the first pass: the frame holds exactly the strata it reports -- the pass of the callback, and whatever the
    # shards' launches (round 6: every shard traces all passes in ONE progressive launch, which does not wait for the callback) had finished by
    # then; shards that stopped behind the furthest one are rendered on to it, so that every pixel holds the same strata
    cam3 = gpu.StaticCamera(200, 100, data.camera, 4, 2, 4)
    assert not ms.render(cam3, progress=lambda c, t: True, samples_per_tick=3)
    n = cam3.currentSample_
    assert n in (3, 6, 8)
    part = gpu.StaticCamera(200, 100, data.camera, 4, 2, 4); part.render(sc, sample_begin=0, sample_end=n)
    assert_same_f32(cam3.acc_, part.acc_, f"cancelled multi render ({n} strata)"); assert (cam3.img_ == part.img_).all()
    # one stratum per pass (the reference's default): still one launch per shard
    seen = []
    cam4 = gpu.StaticCamera(200, 100, data.camera, 4, 2, 4)
    assert ms.render(cam4, progress=lambda c, t: seen.append(c), samples_per_tick=1)
    assert seen == list(range(1, 9))
    # ... and several launches per shard when the records of the range do not fit the cap (1 MB: a few strata of a shard of this frame)
    seen = []
    cam5 = gpu.StaticCamera(200, 100, data.camera, 4, 2, 4)
    assert ms.render(cam5, progress=lambda c, t: seen.append(c), samples_per_tick=1, max_record_mb=1)
    assert seen == list(range(1, 9))
    assert_same_f32(cam5.acc_, ref.acc_, f"{shards} shards, capped records"); assert (cam5.img_ == ref.img_).all()
    assert_same_f32(cam4.acc_, ref.acc_, f"{shards} shards, 8 passes"); assert (cam4.img_ == ref.img_).all()
    ms.destroy()


def test_more_shards_than_tiles(gpu, cornell_pair):
    """A 64x64 frame has four 32x32 tiles: shards 4..7 of 8 own nothing.  launchRender used to divide by their zero
    rad_stride: now an empty shard launches nothing, through jtx_mi_multi_render and through render_shard."""
    import torch
    data, sc, osc = cornell_pair
    ref = gpu.StaticCamera(64, 64, data.camera, 2, 2, 4); ref.render(sc)
    ms = gpu.MultiScene(data, [0] * 8)
    cam = gpu.StaticCamera(64, 64, data.camera, 2, 2, 4)
    assert ms.render(cam)
    assert_same_f32(cam.acc_, ref.acc_, "8 shards, 4 tiles"); assert (cam.img_ == ref.img_).all()
    ms.destroy()
    dev = torch.device("cuda", 0)
    tot = torch.zeros(64 * 64 * 3, dtype=torch.float32, device=dev)
    for r in range(8):
        acc = torch.full((64 * 64 * 3,), 7.0, dtype=torch.float32, device=dev)
        gpu.distributed.render_shard(sc, data.camera_desc(64, 64, 2, 2, 4), r, 8, acc)
        torch.cuda.synchronize()
        if r >= 4:
            assert not acc.any().item(), "a shard without tiles must deliver an all-zero film"
        tot += acc
    assert_same_f32(tot.cpu().numpy().reshape(ref.acc_.shape), ref.acc_, "sum of 8 shards, 4 of them empty")


def test_stale_chunk_counter_does_not_cancel_other_launches(gpu, cornell_pair):
    """after a cancelled persistent pass `last_work` kept pointing at its pushed chunk counter; the next render
    through a launch that owns no counter (count_rays = 1) then read it and returned CANCELLED with an empty film."""
    import threading
    data, sc, osc = cornell_pair
    cam = gpu.StaticCamera(1920, 1080, data.camera, 12, 12, 8)        # one 144-spp pass: ~60 ms on the GPU, so that the timer thread is in time
    cam._pin()                                                        # (page-locking 31 MB of fresh pages takes longer than the timer: do it before the clock runs)
    timer = threading.Timer(0.004, cam.terminateRender)
    timer.start(); cam.render(sc); timer.join()
    if cam.currentSample_ == 144:
        pytest.skip("the frame finished before the cancellation arrived")
    small = gpu.StaticCamera(96, 64, data.camera, 2, 2, 4)
    small.samplesPerPass_ = 2
    seen = []
    small.render(sc, count_rays=True, progress=lambda c, t: seen.append(c))
    ref = gpu.StaticCamera(96, 64, data.camera, 2, 2, 4); ref.render(sc)
    assert seen == [2, 4] and small.currentSample_ == 4
    assert_same_f32(small.acc_, ref.acc_, "counted render after a cancelled persistent one")


def test_cancel_from_the_callback_counts_a_pass_that_completed(gpu, cornell_pair):
    """the pass enqueued before the callback ran may complete before the waves poll the flag (every 64th chunk
    fetch): whatever the race, the film must hold exactly `currentSample_` strata -- never one pass more than reported."""
    data, sc, osc = cornell_pair
    for _ in range(3):
        cam = gpu.StaticCamera(64, 64, data.camera, 4, 2, 4)
        cam.samplesPerPass_ = 2
        cam.render(sc, progress=lambda c, t: cam.terminateRender())         # stop at the first preview; pass 2 is in flight
        n = cam.currentSample_
        assert n in (2, 4, 6, 8)                                            # (up to three passes are in flight behind the callback's)
        part = gpu.StaticCamera(64, 64, data.camera, 4, 2, 4); part.render(sc, sample_begin=0, sample_end=n)
        assert_same_f32(cam.acc_, part.acc_, f"film after a callback cancel at {n}")


def test_two_physical_devices_render_the_one_device_frame(gpu, cornell_pair):
    """MultiScene(data, [0, 1]): two distinct devices, hipMemcpyPeerAsync over xGMI.  Runs wherever >= 2 devices are visible
    (the one-GPU box skips; the same code path runs there with a device listed twice)."""
    import ctypes as C
    lib = gpu._capi.load()
    n = C.c_int32(0); gpu._capi.check(lib.jtx_mi_device_count(C.byref(n)))
    if n.value < 2:
        pytest.skip("one visible device")
    data, sc, osc = cornell_pair
    ref = gpu.StaticCamera(200, 100, data.camera, 4, 2, 4); ref.render(sc)
    ms = gpu.MultiScene(data, list(range(min(n.value, 8))))
    cam = gpu.StaticCamera(200, 100, data.camera, 4, 2, 4)
    assert ms.render(cam)
    assert_same_f32(cam.acc_, ref.acc_, "distinct devices"); assert (cam.img_ == ref.img_).all()
    ms.destroy()


def test_multi_device_full_size_config2(gpu, cornell_pair):
    """C2 (1920x1080, 64 spp) over 8 shards through jtx_mi_multi_render = the 1-device timed frame, bit for bit."""
    data, sc, osc = cornell_pair
    ref = gpu.StaticCamera(1920, 1080, data.camera, 8, 8, 8); ref.render(sc)
    ms = gpu.MultiScene(data, [0] * 8)
    cam = gpu.StaticCamera(1920, 1080, data.camera, 8, 8, 8)
    assert ms.render(cam)
    assert_same_f32(cam.acc_, ref.acc_, "8 shards, C2"); assert (cam.img_ == ref.img_).all()
    ms.destroy()


# ---- integrate / integrateBasic, emission, ThinDielectricBxDF (k_render_alt) ----
@pytest.fixture(scope="module")
def emissive_pair(gpu):
    data = gpu.scenes.emissive()
    sc = gpu.Scene(data); sc.buildBVH()
    yield data, sc, ol.OracleScene(data)
    sc.destroy()


@pytest.mark.parametrize("li", [0, 1, 2])
def test_alternate_integrators_render_like_the_oracle(gpu, emissive_pair, li):
    """integrateMIS with every BxDF incl. THIN_DIELECTRIC (li 0), integrate (li 1: NEE without MIS, emission and sky only
    after a specular bounce) and integrateBasic (li 2: emission at every hit): film, RGB8 image and ray counters of a
    whole frame, bit for bit; tile shards and resumed sample ranges add up."""
    data, sc, osc = emissive_pair
    cam = data.camera_desc(160, 96, 3, 2, 6)
    acc, img, cnt = osc.render(cam, path_integrator=li)
    g = gpu.StaticCamera(160, 96, data.camera, 3, 2, 6)
    g.render(sc, count_rays=True, path_integrator=li)
    assert_same_f32(g.acc_, acc, f"li {li} acc"); assert (g.img_ == img).all(); assert g.counters == cnt
    g.render(sc, count_rays=False, path_integrator=li)
    assert_same_f32(g.acc_, acc, f"li {li} uncounted acc"); assert (g.img_ == img).all()
    assert np.isfinite(acc).all() and acc.max() > 0
    if li == 2:
        assert cnt["n_any"] == 0                                     # integrateBasic never samples a light
    total = np.zeros_like(acc)
    for r in range(3):
        c = gpu.StaticCamera(160, 96, data.camera, 3, 2, 6); c.render(sc, tile_rank=r, tile_world=3, path_integrator=li)
        total += c.acc_
    assert_same_f32(total, acc, "sum of shards")
    part = gpu.StaticCamera(160, 96, data.camera, 3, 2, 6)
    part.render(sc, sample_begin=0, sample_end=2, path_integrator=li); part.render(sc, sample_begin=2, sample_end=6, path_integrator=li)
    assert_same_f32(part.acc_, acc, "resumed")


@pytest.mark.parametrize("li", [0, 1, 2])
def test_alternate_integrators_radiance_samples(gpu, emissive_pair, li):
    data, sc, osc = emissive_pair
    cam = data.camera_desc(320, 200, 4, 4, 8)
    rs = np.random.RandomState(40 + li)
    n = 6000
    rows, cols, ss = rs.randint(0, 200, n), rs.randint(0, 320, n), rs.randint(0, 16, n)
    want = osc.radiance_samples(cam, rows, cols, ss, path_integrator=li)
    got = gpu.api.radiance_samples(sc, cam, rows, cols, ss, path_integrator=li)
    assert_same_f32(got, want, f"li {li} per-sample radiance")
    assert (want > 0).any()


def test_the_three_integrators_differ_where_they_should(gpu, emissive_pair):
    """integrateMIS ignores emission (integrator.cpp:189-190); integrateBasic sees the emitters but no point light."""
    data, sc, osc = emissive_pair
    f = {}
    for li in (0, 1, 2):
        g = gpu.StaticCamera(96, 64, data.camera, 2, 2, 4); g.render(sc, path_integrator=li); f[li] = g.acc_.copy()
    assert not np.array_equal(f[0], f[1]) and not np.array_equal(f[1], f[2]) and not np.array_equal(f[0], f[2])


def test_thin_dielectric_bxdf_batch(gpu, emissive_pair):
    """ThinDielectricBxDF through the BxDF batch entry points: sample bit-equal to the oracle, eval = 0, pdf = 0."""
    data, sc, osc = emissive_pair
    thin = [i for i, m in enumerate(data.materials) if m["type"] == gpu.scenes.THIN_DIELECTRIC]
    assert len(thin) == 2
    nrm, wo, wi, uc, u2, uv = _bxdf_inputs(20000, 77)
    for mi in thin:
        a = osc.sampleBxdf(mi, nrm, wo, uc, u2, uv)
        b = sc.sampleBxdf(mi, nrm, wo, uc, u2, uv)
        assert (a["ok"] == b["ok"]).all() and a["ok"].mean() > 0.5
        for k in ("f", "wi", "pdf"):
            assert_same_f32(b[k], a[k], f"thin dielectric sample {k}")
        assert not sc.evalBxdf(mi, nrm, wo, wi, uv).any() and not sc.pdfBxdf(mi, nrm, wo, wi, uv).any()
        assert not osc.evalBxdf(mi, nrm, wo, wi, uv).any() and not osc.pdfBxdf(mi, nrm, wo, wi, uv).any()


# ---- device refit after transform edits (jtx_refit.hip) ----
def _edit(gpu, data, seed):
    """a translate + non-uniform scale + rotation for a few meshes (what Display's edit panel produces, display.cpp:545-588)"""
    rs = np.random.RandomState(seed)
    out = {}
    for mi in rs.choice(len(data.meshes), min(4, len(data.meshes)), replace=False):
        a = rs.uniform(-0.4, 0.4)
        rot = np.array([[np.cos(a), 0, np.sin(a), 0], [0, 1, 0, 0], [-np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]], np.float32)
        sc = np.diag(list(rs.uniform(0.8, 1.25, 3)) + [1.0]).astype(np.float32)
        tr = np.eye(4, dtype=np.float32); tr[:3, 3] = rs.uniform(-25, 25, 3)
        out[int(mi)] = (tr @ rot @ sc).astype(np.float32)
    return out


def _same_tree(n0, r0, n1, r1, what):
    """node for node: boxes, child links / leaf ranges, split axes; and Scene::triangles_ in the same order"""
    assert len(n0) == len(n1), f"{what}: {len(n0)} vs {len(n1)} nodes"
    assert (n0["pmin"] == n1["pmin"]).all() and (n0["pmax"] == n1["pmax"]).all(), f"{what}: boxes differ"
    assert (n0["offset"] == n1["offset"]).all() and (n0["num_prims"] == n1["num_prims"]).all(), f"{what}: links / leaf ranges differ"
    inner = n0["num_prims"] == 0
    assert (n0["axis"][inner] == n1["axis"][inner]).all(), f"{what}: split axes differ"
    assert (r0["index"] == r1["index"]).all() and (r0["mesh_index"] == r1["mesh_index"]).all(), f"{what}: primitive order differs"


@pytest.mark.parametrize("which,max_prims", [("cornell", 1), ("mixed", 1), ("atrium", 1), ("atrium", 4), ("quads", 2)])
def test_device_rebuild_gives_the_reference_tree(gpu, which, max_prims):
    """jtx_mi_scene_rebuild (Scene::rebuildBVH on the device, csrc/jtx_build_dev.hip): (a) rebuilding an unedited
    scene gives the host builder's tree -- the reference's (bvh.cpp:9-149) -- node for node AND primitive for primitive
    (std::partition's order included: the atrium keeps its quads in two-primitive leaves), so the frame (film, RGB8, all nine
    ray counters: same node visits, same triangle tests) equals the oracle's bit for bit; (b) after transform edits the rebuilt
    tree equals a FRESH host build of the edited scene and the frame the oracle's frame of the edited scene; (c) a refit
    after a rebuild works on the new topology."""
    make = {"cornell": gpu.scenes.cornell, "mixed": lambda: gpu.scenes.mixed(sphere_res=(16, 8)),
            "atrium": lambda: gpu.scenes.atrium(target_tris=30000),
            "quads": lambda: gpu.scenes.atrium(target_tris=6000)}[which]
    data = make(); data.max_prims_in_node = max_prims
    sc = gpu.Scene(data); sc.buildBVH(max_prims)
    n0, r0 = sc.bvh()
    assert which == "mixed" or (n0["num_prims"] > 1).any()             # leaves of several primitives: their inner order matters
    W, H = 200, 120
    osc = ol.OracleScene(data)
    cam = data.camera_desc(W, H, 2, 2, 5)
    acc, img, cnt = osc.render(cam)
    if which == "atrium":
        sc.reserveRebuild(); sc.reserveRebuild()                        # the edit loop's memory ahead of the first edit (idempotent)
    sc.rebuildBVHOnDevice(max_prims)
    info = sc.info()
    assert info["device_built"] and not info["refitted"]
    n1, r1 = sc.bvh()
    _same_tree(n0, r0, n1, r1, f"{which} rebuilt")
    for count in (True, False):
        g = gpu.StaticCamera(W, H, data.camera, 2, 2, 5); g.render(sc, count_rays=count)
        assert_same_f32(g.acc_, acc, f"{which}: frame on the rebuilt tree (count={count})"); assert (g.img_ == img).all()
        if count:
            assert g.counters == cnt, "same tree => same node visits and triangle tests"
    # (b) edits
    edits = _edit(gpu, data, 7)
    fresh_data = make(); fresh_data.max_prims_in_node = max_prims
    for mi, m in edits.items():
        sc.setTransform(mi, m)
        fresh_data.meshes[mi]["transform"] = m.copy()
    sc.rebuildBVHOnDevice(max_prims)
    fresh = gpu.Scene(fresh_data); fresh.buildBVH(max_prims)
    nf, rf = fresh.bvh()
    n2, r2 = sc.bvh()
    _same_tree(nf, rf, n2, r2, f"{which} edited + rebuilt")
    assert abs(sc.getSceneRadius() - fresh.getSceneRadius()) == 0
    acc2, img2, cnt2 = ol.OracleScene(fresh_data).render(cam)
    for count in (True, False):
        g = gpu.StaticCamera(W, H, data.camera, 2, 2, 5); g.render(sc, count_rays=count)
        assert_same_f32(g.acc_, acc2, f"{which}: edited frame on the rebuilt tree (count={count})"); assert (g.img_ == img2).all()
        if count:
            assert g.counters == cnt2
    # (c) the refit sources follow the rebuild: identity refit on the new topology reproduces the frame
    for mi, m in edits.items():
        sc.setTransform(mi, m)
    sc.refit()
    g = gpu.StaticCamera(W, H, data.camera, 2, 2, 5); g.render(sc, count_rays=False)
    assert_same_f32(g.acc_, acc2, f"{which}: refit after rebuild")
    fresh.destroy(); sc.destroy()


def test_device_rebuild_full_size_atrium(gpu, atrium_full):
    """the 256 k-triangle device rebuild under test (what tools/rebuild_time.py prints, asserted): the tree of
    jtx_mi_scene_rebuild equals the host build's node for node and primitive for primitive (283 657 nodes, depth 22, 114 k
    two-primitive leaves), and the uncounted 480x270x16 frame on the device-built structures -- their 8-ary nodes are laid out level
    by level instead of depth first -- equals the host-built scene's bit for bit; a second rebuild (the steady state of an edit
    loop: no allocation) gives the same again."""
    data, sc, osc = atrium_full
    fresh = gpu.Scene(data); fresh.buildBVH()
    try:
        n0, r0 = sc.bvh()
        W, H = 480, 270
        ref = gpu.StaticCamera(W, H, data.camera, 4, 4, 8); ref.render(sc, count_rays=False)
        for again in range(2):
            fresh.rebuildBVHOnDevice()
            info = fresh.info()
            assert info["device_built"] and not info["refitted"] and info["wide_depth"] >= 2
            n1, r1 = fresh.bvh()
            _same_tree(n0, r0, n1, r1, f"atrium_full rebuilt ({again})")
            g = gpu.StaticCamera(W, H, data.camera, 4, 4, 8); g.render(fresh, count_rays=False)
            assert_same_f32(g.acc_, ref.acc_, f"atrium_full: frame on the device-built structures ({again})"); assert (g.img_ == ref.img_).all()
    finally:
        fresh.destroy()


_FAILURE_ATOMIC_CHILD = r'''
import os, sys
import numpy as np
import torch
torch.zeros(1, device="cuda")
import jtx_pathtracer_amd as gpu
assert os.path.basename(gpu._capi.load()._name) == "libjtx_mi_testhooks.so"
data = gpu.scenes.atrium(target_tris=12000)
sc = gpu.Scene(data); sc.buildBVH()
def same_tree(n0, r0, n1, r1):
    inner = n0["num_prims"] == 0
    return (len(n0) == len(n1) and (n0["pmin"] == n1["pmin"]).all() and (n0["pmax"] == n1["pmax"]).all() and (n0["offset"] == n1["offset"]).all()
            and (n0["num_prims"] == n1["num_prims"]).all() and (n0["axis"][inner] == n1["axis"][inner]).all()
            and (r0["index"] == r1["index"]).all() and (r0["mesh_index"] == r1["mesh_index"]).all())
n0, r0 = sc.bvh()
ref = gpu.StaticCamera(160, 90, data.camera, 2, 2, 5); ref.render(sc, count_rays=False)
m = np.eye(4, dtype=np.float32); m[0, 3] = 2.5
sc.setTransform(0, m)
os.environ["JTX_FAIL_REBUILD_BEFORE_COMMIT"] = "1"
try:
    sc.rebuildBVHOnDevice()
    print("FAIL: the injected failure did not fire"); sys.exit(1)
except gpu._capi.JtxMiError as e:
    assert "injected failure" in str(e), str(e)
del os.environ["JTX_FAIL_REBUILD_BEFORE_COMMIT"]
n1, r1 = sc.bvh()
assert same_tree(n0, r0, n1, r1), "tree changed by the failed rebuild"
assert not sc.info()["device_built"]
g = gpu.StaticCamera(160, 90, data.camera, 2, 2, 5); g.render(sc, count_rays=False)
assert np.array_equal(np.asarray(g.acc_).view(np.uint32), np.asarray(ref.acc_).view(np.uint32)), "frame after the failed rebuild"
sc.rebuildBVHOnDevice()                                       # the edit is still pending: now it takes effect
fresh_data = gpu.scenes.atrium(target_tris=12000); fresh_data.meshes[0]["transform"] = m.copy()
fresh = gpu.Scene(fresh_data); fresh.buildBVH()
nf, rf = fresh.bvh(); n2, r2 = sc.bvh()
assert same_tree(nf, rf, n2, r2), "rebuild after the failed one"
print("failure-atomic: ok")
'''


def test_device_rebuild_is_failure_atomic(gpu, tmp_path):
    """a rebuild that fails late -- right before its commit section, after every kernel has run and every new buffer
    has been written -- leaves the scene exactly as it was: same tree, same frame, and the next rebuild works.  The fault is injected by
    a hook that only libjtx_mi_testhooks.so carries (jtx_capi.hip -DJTX_TEST_HOOKS; ADVICE r4: the product library has none), so the
    scenario runs in a child process that loads that library through JTX_MI_LIB; the product library ignores the variable."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "jtx-pathtracer_amd", "libjtx_mi_testhooks.so")
    assert os.path.exists(lib), "build it with __graft_entry__.build() (jtx.build_test_hooks)"
    script = tmp_path / "failure_atomic_child.py"
    script.write_text(_FAILURE_ATOMIC_CHILD)
    env = dict(os.environ, JTX_MI_LIB=lib, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "failure-atomic: ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    # the product library has no such hook: the variable changes nothing
    data = gpu.scenes.atrium(target_tris=3000)
    sc = gpu.Scene(data); sc.buildBVH()
    os.environ["JTX_FAIL_REBUILD_BEFORE_COMMIT"] = "1"
    try:
        sc.rebuildBVHOnDevice()
        assert sc.info()["device_built"]
    finally:
        del os.environ["JTX_FAIL_REBUILD_BEFORE_COMMIT"]
        sc.destroy()


def test_device_rebuild_declines_what_the_reference_cannot_build(gpu):
    """the device builder says WHY it does not build (it used to return hipErrorUnknown) -- here an edit that scales a mesh
    until its boxes' surface areas overflow fp32: every SAH cost is inf, the reference's buildTree would recurse for ever (bvh.cpp:96-111,
    126-127) -- and the scene stays as it was."""
    data = gpu.scenes.atrium(target_tris=4000)
    sc = gpu.Scene(data); sc.buildBVH()
    try:
        n0, r0 = sc.bvh()
        sc.setTransform(0, np.diag([1e25, 1e25, 1e25, 1.0]).astype(np.float32))
        with pytest.raises(gpu._capi.JtxMiError, match="declined.*does not terminate"):
            sc.rebuildBVHOnDevice()
        n1, r1 = sc.bvh()
        _same_tree(n0, r0, n1, r1, "after the declined rebuild")
        sc.setTransform(0, np.eye(4, dtype=np.float32))
        sc.rebuildBVHOnDevice()
        n2, r2 = sc.bvh()
        _same_tree(n0, r0, n2, r2, "rebuild after the edit was taken back")
    finally:
        sc.destroy()


def test_device_rebuild_edge_cases(gpu):
    """the corners of buildTree (bvh.cpp:18-57) through jtx_mi_scene_rebuild: one primitive (a leaf root), a quad whose halves
    share their centroid (degenerate centroid bounds: one leaf of two), two separate triangles in either input order
    (nth_element swaps or keeps), three and five primitives (buckets with a single primitive, empty buckets in between), a
    flat cloud (zero surface area: a leaf whatever its size)"""
    sc_ = gpu.scenes
    def tri(x, y, z, s=1.0):
        return np.array([[x, y, z], [x + s, y, z], [x, y + s, z]], np.float32)
    def scene(tris, flat=False):
        d = sc_.SceneData("edge")
        d.materials = [sc_.material(sc_.DIFFUSE, (0.7, 0.7, 0.7))]
        v = np.concatenate(tris).astype(np.float32)
        d.add_mesh(np.arange(len(v), dtype=np.int32).reshape(-1, 3), v, np.tile(np.array([[0, 0, 1]], np.float32), (len(v), 1)), 0)
        d.lights = [sc_.light(sc_.POINT, (0.0, 0.0, 5.0), (1, 1, 1), 10.0)]
        d.camera = dict(center=(0.0, 0.0, 9.0), target=(0.0, 0.0, 0.0), up=(0, 1, 0), yfov=40.0, defocus_angle=0.0, focus_distance=1.0)
        return d
    quad = [np.array([[-1, -1, 0], [-1, 1, 0], [1, 1, 0]], np.float32), np.array([[-1, -1, 0], [1, 1, 0], [1, -1, 0]], np.float32)]
    cases = {
        "one": [tri(0, 0, 0)],
        "quad": quad,
        "two, in order": [tri(-2, 0, 0), tri(1, 0, 0)],
        "two, swapped": [tri(1, 0, 0), tri(-2, 0, 0)],
        "three": [tri(1, 0, 0), tri(-2, 0, 0), tri(0, 1.5, -1)],
        "five on a line": [tri(float(x), 0, 0, 0.5) for x in (3, -4, 0, 1, -1)],
        "degenerate points": [np.zeros((3, 3), np.float32) + np.float32(k) for k in (0, 0, 1)],
    }
    for name, tris in cases.items():
        data = scene(tris)
        sc = gpu.Scene(data); sc.buildBVH()
        n0, r0 = sc.bvh()
        sc.rebuildBVHOnDevice()
        n1, r1 = sc.bvh()
        _same_tree(n0, r0, n1, r1, name)
        assert sc.info()["max_depth"] == gpu.api.bvh_build_host(data)[2]
        ref = gpu.Scene(data); ref.buildBVH()
        for count in (True, False):
            a = gpu.StaticCamera(48, 48, data.camera, 2, 2, 3); a.render(sc, count_rays=count)
            b = gpu.StaticCamera(48, 48, data.camera, 2, 2, 3); b.render(ref, count_rays=count)
            assert_same_f32(a.acc_, b.acc_, f"{name}: frame after a device rebuild (count={count})")
            if count:
                assert a.counters == b.counters
        sc.destroy(); ref.destroy()


@pytest.mark.parametrize("which", ["cornell", "mixed", "atrium"])
def test_refit_equals_a_fresh_build_of_the_edited_scene(gpu, which):
    """jtx_mi_scene_refit: (a) setting the SAME transforms again reproduces the built scene bit for bit (film and ray
    counters); (b) after real edits every ray finds the hit distance a FRESH build of the edited scene finds (binary records
    and 8-ary nodes), the same primitive except among equal distances, and the rendered frame equals the oracle's frame of
    the edited scene on all but a handful of tie pixels; (c) the refitted boxes are the boxes a build computes."""
    make = {"cornell": gpu.scenes.cornell, "mixed": lambda: gpu.scenes.mixed(sphere_res=(16, 8)),
            "atrium": lambda: gpu.scenes.atrium(target_tris=30000)}[which]
    data = make()
    sc = gpu.Scene(data); sc.buildBVH()
    W, H = 200, 120
    ref = gpu.StaticCamera(W, H, data.camera, 2, 2, 5); ref.render(sc, count_rays=True)
    base_cnt = dict(ref.counters)
    # (a) identity edit
    for mi in range(len(data.meshes)):
        sc.setTransform(mi, data.meshes[mi]["transform"])
    sc.refit()
    assert sc.info()["refitted"]
    again = gpu.StaticCamera(W, H, data.camera, 2, 2, 5); again.render(sc, count_rays=True)
    assert_same_f32(again.acc_, ref.acc_, "identity refit"); assert again.counters == base_cnt
    again.render(sc, count_rays=False)
    assert_same_f32(again.acc_, ref.acc_, "identity refit, uncounted")
    # (b) real edits
    edits = _edit(gpu, data, 5)
    fresh_data = make()
    for mi, m in edits.items():
        sc.setTransform(mi, m)
        fresh_data.meshes[mi]["transform"] = m.copy()
    sc.refit()
    fresh = gpu.Scene(fresh_data); fresh.buildBVH()
    osc = ol.OracleScene(fresh_data)
    o, d = _rays_for(fresh_data, osc, 60000, 9)
    a, b = sc.closestHit(o, d), fresh.closestHit(o, d)
    assert (a["hit"] == b["hit"]).all() and a["hit"].mean() > 0.3
    assert_same_f32(a["t"], b["t"], "hit distance after refit vs fresh build")
    same_prim = (a["point"] == b["point"]).all(axis=1) & (a["normal"] == b["normal"]).all(axis=1)
    assert same_prim.mean() > 0.9995                                  # differences only among exactly equal distances
    tmax = np.where(a["hit"] > 0, a["t"] * 0.999, 1e30).astype(np.float32)
    assert (sc.anyHit(o, d, 0.0, tmax) == fresh.anyHit(o, d, 0.0, tmax)).all()
    cam = fresh_data.camera_desc(W, H, 2, 2, 5)
    acc, img, _ = osc.render(cam, count=False)
    for count in (True, False):                                      # binary records, then the wide nodes / LDS copy
        g = gpu.StaticCamera(W, H, data.camera, 2, 2, 5); g.render(sc, count_rays=count)
        diff = (g.acc_.view(np.uint32) != acc.view(np.uint32)).any(axis=2)
        assert diff.mean() < 2e-3, f"{which}: {diff.sum()} pixels differ after refit (count={count})"
    # (c) boxes: every refitted node box = union of its primitives' boxes of the edited scene (what a build computes)
    nodes, refs = sc.bvh()
    tri = np.stack([_tri_world(fresh_data, r) for r in refs[:: max(1, len(refs) // 2000)]])
    leaves = nodes[nodes["num_prims"] > 0]
    assert np.isfinite(nodes["pmin"]).all() and (nodes["pmin"] <= nodes["pmax"]).all()
    root = nodes[0]
    allv = np.concatenate([_tri_world(fresh_data, r) for r in refs])
    assert np.array_equal(root["pmin"], allv.min(0)) and np.array_equal(root["pmax"], allv.max(0))
    fr_nodes, _ = fresh.bvh()
    assert np.array_equal(fr_nodes[0]["pmin"], root["pmin"]) and np.array_equal(fr_nodes[0]["pmax"], root["pmax"])
    assert abs(sc.info()["scene_radius"] - fresh.info()["scene_radius"]) == 0
    sc.destroy(); fresh.destroy()


def _tri_world(data, ref):
    idx, mesh = int(ref[0]), int(ref[1])
    m = data.meshes[mesh]
    T = m["transform"].astype(np.float32)
    v = m["vertices"][m["indices"][idx]].astype(np.float32)          # (3, 3)
    out = np.empty((3, 3), np.float32)
    for r in range(3):                                               # applyToPoint row by row, fp32 left to right
        out[:, r] = ((T[r, 0] * v[:, 0] + T[r, 1] * v[:, 1]) + T[r, 2] * v[:, 2]) + T[r, 3]
    return out


def test_closed_form_films_on_the_gpu(gpu):
    """the two closed-form scenes of tests/test_geometry_physics_cpu.py rendered by the HIP kernels (timed kernel and the
    counting one): the film must equal rho * sky over the convex Lambertian body, and the float64 direct-lighting formula on
    the floor under the point light -- a check against physics, not against the twin restatement"""
    import physics_cases as pc
    s = pc.furnace_scene()
    sc = gpu.Scene(s); sc.buildBVH()
    for count in (False, True):
        cam = gpu.StaticCamera(96, 96, s.camera, 2, 2, 8)
        cam.render(sc, count_rays=count, integrator=1)
        pc.check_furnace_film(np.asarray(cam.acc_, np.float32), 96, 96, 4)
    s = pc.plane_scene()
    sc = gpu.Scene(s); sc.buildBVH()
    W, H = 64, 48
    camd = s.camera_desc(W, H, 2, 2, 1)
    rows, cols, smp = np.meshgrid(np.arange(H), np.arange(W), np.arange(4), indexing="ij")
    ro, rd = ol.camera_rays(camd, rows.ravel(), cols.ravel(), smp.ravel())
    want = pc.plane_film(ro, rd, W, H, 4)
    for count in (False, True):
        cam = gpu.StaticCamera(W, H, s.camera, 2, 2, 1)
        cam.render(sc, count_rays=count, integrator=1)
        assert np.allclose(np.asarray(cam.acc_, np.float32).reshape(H, W, 3), want, rtol=3e-5, atol=1e-6)


def test_obj_with_an_exr_diffuse_map_renders_like_the_oracle(gpu, tmp_path):
    """ingestion end to end: an OBJ whose .mtl names an EXR map (loader.cpp:64-101 -> TextureImage::load -> tinyexr: RGBA
    floats, channels_ = 4) and a JPEG map (stbi_loadf, 3 channels), loaded by scenes.load_obj, rendered by the HIP kernels
    and by the oracle: bit-identical film"""
    from jtx_pathtracer_amd import gltf
    rs = np.random.RandomState(4)
    (tmp_path / "maps").mkdir()
    yy, xx = np.mgrid[0:32, 0:48]
    img = np.stack([0.2 + 0.7 * ((xx // 6 + yy // 4) % 2), 0.5 + 0.4 * np.sin(xx / 5.0), 0.3 + 0.02 * yy], -1).astype(np.float32)
    (tmp_path / "maps" / "floor.exr").write_bytes(gltf.encode_exr(img, compression="zip"))
    from PIL import Image
    Image.fromarray((rs.rand(24, 24, 3) * 255).astype(np.uint8)).resize((96, 96), 0).save(str(tmp_path / "maps" / "wall.jpg"), quality=92)
    (tmp_path / "room.mtl").write_text("newmtl floor\nmap_Kd maps/floor.exr\nnewmtl wall\nmap_Kd maps/wall.jpg\n")
    (tmp_path / "room.obj").write_text(
        "mtllib room.mtl\nv -2 0 -2\nv 2 0 -2\nv 2 0 2\nv -2 0 2\nv -2 3 -2\nv 2 3 -2\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 1 0\nvn 0 0 1\n"
        "o floor\nusemtl floor\nf 1/1/1 4/4/1 3/3/1 2/2/1\no wall\nusemtl wall\nf 1/1/2 2/2/2 6/3/2 5/4/2\n")
    white = gpu.scenes.material(gpu.scenes.DIFFUSE, (1.0, 1.0, 1.0))
    s = gpu.scenes.load_obj(str(tmp_path / "room.obj"), default_material=white)
    assert len(s.textures) == 2 and s.textures[0].shape == (32, 48, 4) and s.textures[1].shape == (96, 96, 3)
    s.lights = [gpu.scenes.light(gpu.scenes.POINT, (0.5, 2.5, 1.0), (1, 1, 1), 12.0)]
    s.sky = (0.2, 0.3, 0.4)
    s.camera = dict(center=(0.0, 2.0, 5.0), target=(0.0, 1.0, 0.0), up=(0, 1, 0), yfov=45.0, defocus_angle=0.0, focus_distance=1.0)
    sc = gpu.Scene(s); sc.buildBVH()
    osc = ol.OracleScene(s)
    acc, img8, cnt = osc.render(s.camera_desc(160, 120, 2, 2, 4))
    for count in (True, False):
        cam = gpu.StaticCamera(160, 120, s.camera, 2, 2, 4)
        cam.render(sc, count_rays=count, integrator=1)
        assert_same_f32(cam.acc_, acc, "OBJ + EXR map")
        assert (cam.img_ == img8).all()
    assert len(np.unique(img8.reshape(-1, 3), axis=0)) > 500          # the maps show


def test_cpp_host_loads_an_asset_and_renders_it(gpu, tmp_path):
    """a C++ host end to end (tests/cpp/host_asset_demo.cpp): loadScene of a GLB with PNG maps (host/jtx_host_loader.hpp) ->
    buildBVH -> StaticCamera::renderFinal -> Camera::save; the image equals the Python mirror's render of the same file"""
    import os, subprocess, zlib
    from jtx_pathtracer_amd import gltf
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "jtx-pathtracer_amd")
    exe = str(tmp_path / "host_asset_demo")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(root, "tests", "cpp", "host_asset_demo.cpp"),
                    "-L", libdir, "-ljtx_mi", "-lpthread", "-Wl,-rpath," + libdir], check=True)
    data = gpu.scenes.mixed(sphere_res=(12, 6), textured=True)
    tex8 = [(np.clip(np.asarray(t)[..., :3], 0, 1) ** (1 / 2.2) * 255).astype(np.uint8) for t in data.textures]
    glb = str(tmp_path / "room.glb")
    gltf.write_glb(glb, data, textures_u8=tex8)
    png = str(tmp_path / "out.png")
    out = subprocess.run([exe, glb, png], check=True, capture_output=True, text=True).stdout.split()
    kv = dict(zip(out[0::2], out[1::2]))
    s = gltf.load_gltf(glb)
    s.sky = (0.5, 0.7, 1.0)
    s.lights = [gpu.scenes.light(gpu.scenes.POINT, (278.0, 500.0, 279.5), (1, 1, 1), 60000.0)]
    s.camera = dict(center=(278.0, 273.0, -800.0), target=(278.0, 273.0, 0.0), up=(0, 1, 0), yfov=39.3077, defocus_angle=0.0, focus_distance=1.0)
    assert int(kv["meshes"]) == len(s.meshes) and int(kv["triangles"]) == s.num_triangles and int(kv["textures"]) == 2
    sc = gpu.Scene(s); sc.buildBVH()
    cam = gpu.StaticCamera(96, 72, s.camera, 2, 2, 5)
    cam.render(sc, count_rays=False)
    assert kv["crc"] == "%08x" % zlib.crc32(np.ascontiguousarray(cam.img_).tobytes())
    assert np.array_equal(gltf.decode_png(open(png, "rb").read()), np.asarray(cam.img_).reshape(72, 96, 3)[::-1])
    assert len(np.unique(np.asarray(cam.img_).reshape(-1, 3), axis=0)) > 8          # (every glTF material is metal-rough white: mostly mirrored sky)


def test_bench_times_a_scene_file_and_the_extra_workload_path(gpu, tmp_path):
    """bench.py --scene (VERDICT r3 missing 4) and the per-workload timing the default run adds to its line (next 2), on the device: an OBJ
    of the Cornell geometry goes through createScene's rules, renders at the headline's frame size and reports rays, kernel time and the
    counter-free roofline fraction (an OBJ brings no lights: extension rays only, 0.43 G per frame)."""
    import torch
    import bench
    p = os.path.join(tmp_path, "room.obj")
    gpu.scenes.write_obj(gpu.scenes.cornell(), p)
    name, data, dims = bench.load_workload(gpu, "cornell_1920x1080_64spp_d8", scene_file=p, camera="278,273,-800,278,273,0,39.3077")
    assert name == "file:room_1920x1080_64spp_d8" and dims == (1920, 1080, 8, 8, 8)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        e = bench.time_workload(gpu, torch, dev, st, name, data, dims, steps=2, warmup=1)
    # (kernel_ms: launches with one frame in flight; ms_per_step: the pipelined loop, resolve pass included -- faster per frame than a lone launch + resolve)
    assert e["rays_per_frame"] > 3e8 and e["scene_triangles"] == 32 and e["kernel_ms"] > 1.0 and e["frames_in_flight"] == 3
    # (two frames are no steady state, and the suite's other tests leave the card in whatever clock / memory state they left it: a sanity
    #  bound, not a performance gate -- that is bench.py's recorded line)
    assert e["ms_per_step"] <= e["kernel_ms"] * 1.5 + 1.0 and e["in_flight"]["useful_frac"] >= e["useful_frac"] * 0.6
    assert abs(e["value"] - e["rays_per_frame"] / (e["ms_per_step"] * 1e-3) / 1e6) / e["value"] < 1e-3
    assert 0.05 < e["useful_frac"] < 0.4 and e["frac"] is None             # a file scene has no recorded counters: only the counter-free fraction


def test_rccl_takes_the_exchange_of_the_sharded_frame(gpu):
    """The N > 1 path's collectives on REAL RCCL (the gloo rehearsals cannot touch it): a one-GPU box offers one rank, so
    the group has one member and the gather / reduce degenerate to self-copies -- through RCCL's communicator set-up and torch's c10d
    checks for exactly the slab shapes, dtypes and views `distributed.FrameGather` / `reduce_frame` / bench.py use.  In a child
    process (tools/rccl_selfcheck.py) with a time limit, so that a communicator that cannot come up costs a failure, not the session."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_selfcheck.py"), "--width", "640", "--height", "360"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["rccl_selfcheck"] == "ok" and line["gather"] and line["reduce"] and line["slab_bytes"] == 640 * 360 * 15


def test_device_built_and_refitted_wide_nodes_walk_like_the_binary_tree(gpu):
    """The 8-ary node set as it stands ON THE DEVICE (jtx_mi_scene_get_wide) after a device rebuild and after a refit, walked by the
    CPU test's walker (tests/test_wide_bvh_cpu.py: its decoder asserts, per node and octant, one-hot positions == visiting order and
    octant 7 - q == octant q reversed): for random rays it reaches the leaves the reference's binary traversal reaches, in the
    reference's order -- through the quantised nodes and through the root-peel record.  The host builder's array has this test on the
    CPU; the device builder (k_wide_fill) and the refit kernels write the same layout through the same encoders, and this is where
    their OUTPUT is read back and checked word by word rather than through a film."""
    import test_wide_bvh_cpu as W
    rs = np.random.RandomState(11)

    def check(sc, what, rays=40):
        nodes, _ = sc.bvh()
        w = sc.wide()
        assert len(w) == sc.info()["wide_bytes"] // 16 and len(w) > W.FIRST_BLOCK
        lo, hi = nodes[0]["pmin"], nodes[0]["pmax"]
        nonempty = 0
        for i in range(rays):
            o = (lo + (hi - lo) * rs.uniform(0.05, 0.95, 3)).astype(np.float32)
            d = rs.normal(size=3).astype(np.float32)
            inv = (np.float32(1.0) / d).astype(np.float32)
            neg = [int(inv[k] < 0) for k in range(3)]
            tmax = np.float32(np.inf) if i % 3 == 0 else np.float32(rs.uniform(1.0, 60.0))
            a = W.binary_leaves(nodes, o, inv, neg, np.float32(0.001), tmax)
            assert W.wide_leaves(w, o, inv, neg, np.float32(0.001), tmax) == a, f"{what}: ray {i}"
            assert W.wide_leaves_peeled(w, o, inv, neg, np.float32(0.001), tmax) == a, f"{what}: ray {i} (root peel)"
            nonempty += bool(a)
        assert nonempty > rays // 3, what

    data = gpu.scenes.atrium(target_tris=3000)
    sc = gpu.Scene(data); sc.buildBVH()
    check(sc, "host-built, as uploaded")
    sc.rebuildBVHOnDevice(1)
    assert sc.info()["device_built"]
    check(sc, "device-built")
    for mi, m in _edit(gpu, data, 3).items():
        sc.setTransform(mi, m)
    sc.rebuildBVHOnDevice(1)
    check(sc, "edited + device-built")
    for mi, m in _edit(gpu, data, 5).items():
        sc.setTransform(mi, m)
    sc.refit()
    assert sc.info()["refitted"]
    check(sc, "refitted")
    sc.destroy()


def test_bench_starts_its_own_ranks(gpu):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE (the shape of the driver's N = 1 command, VERDICT r4 missing 3):
    the parent -- which has not touched the GPU -- starts torch.distributed.run as a child and relays rank 0's one JSON line and the
    return code.  Rehearsal form (both ranks on this card, gloo through the host): `nranks_seen` 2, the shard counters add up to the
    one-GPU frame's ray count (5.49 rays per sample)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(JTX_DIST_BACKEND="gloo", JTX_ALL_RANKS_ON_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"]["nranks_seen"] == 2 and "REHEARSAL" in line["config"]["parallelism"]
    assert sum(x["shard_rays"] for x in line["ranks"]["ranks"]) == line["config"]["rays_per_frame"] == 727984390


def test_bench_self_launch_with_rccl_at_one_rank(gpu):
    """the path the driver's multi-GPU run takes -- bench.py starting its own ranks as a torch.distributed.run child, an
    `nccl` (= RCCL) process group, the per-frame gather through RCCL overlapped with the next frame, host delivery on the destination rank,
    the rank diagnosis -- with ONE rank (JTX_BENCH_FORCE_DIST=1), which is what a one-GPU box can run of it for real (the two-rank
    rehearsal above goes over gloo)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "JTX_DIST_BACKEND", "JTX_ALL_RANKS_ON_DEVICE")}
    env.update(JTX_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["ranks"]["nranks_seen"] == 1 and "REHEARSAL" not in line["config"]["parallelism"]
    assert "gather/frame overlapped" in line["config"]["parallelism"] and "delivered to page-locked host memory" in line["config"]["timed_region"]
    assert line["ranks"]["ranks"][0]["shard_rays"] == line["config"]["rays_per_frame"] == 727984390
    assert line["ranks"]["ranks"][0]["exchange_ms"] is not None and 20.0 < line["ms_per_step"] < 40.0


_REFUSED_PEER_CHILD = r"""
import numpy as np
import jtx_pathtracer_amd as gpu
gpu._capi.check(gpu._capi.load().jtx_mi_set_device(0))
data = gpu.scenes.cornell()
sc = gpu.Scene(data); sc.buildBVH()
ref = gpu.StaticCamera(200, 100, data.camera, 4, 2, 4); ref.render(sc)
ms = gpu.MultiScene(data, [0, 0, 0])
cam = gpu.StaticCamera(200, 100, data.camera, 4, 2, 4)
seen = []
assert ms.render(cam, progress=lambda c, t: seen.append(c), samples_per_tick=3)
assert seen == [3, 6, 8]
assert np.array_equal(cam.acc_.view(np.uint32), ref.acc_.view(np.uint32)) and np.array_equal(cam.img_, ref.img_)
ms.destroy(); sc.destroy()
print("refused-peer: ok")
"""


def test_multi_device_exchange_when_peer_access_is_refused(gpu, tmp_path):
    """jtx_mi_multi_create on a platform that refuses peer access (hipDeviceCanAccessPeer -> 0): the slabs then travel
    through the library's page-locked stage (device -> host on the shard's stream, host -> device 0 behind an event) instead of
    hipMemcpyPeerAsync -- a branch no box of this pool takes by itself.  The refusal is injected by a hook that only
    libjtx_mi_testhooks.so carries (jtx_multi.hip -DJTX_TEST_HOOKS), in a child process; three shards, previews per pass, the one-device
    frame bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "jtx-pathtracer_amd", "libjtx_mi_testhooks.so")
    assert os.path.exists(lib), "build it with __graft_entry__.build() (jtx.build_test_hooks)"
    script = tmp_path / "refused_peer_child.py"
    script.write_text(_REFUSED_PEER_CHILD)
    env = dict(os.environ, JTX_MI_LIB=lib, JTX_TEST_REFUSE_PEER_ACCESS="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "refused-peer: ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


# ---- per-ray parity THROUGH THE TRAVERSALS THAT SHIP (VERDICT r4 missing 4 / weak 2): jtx_mi_closest_hit_batch_via / any_hit_batch_via
# with JTX_MI_TRAVERSAL_PRODUCTION run traverseLeaves (Cornell: the flat leaf list, scalar-operand boxes), the LDS copy (scenes that
# are LDS-resident but have more than 32 leaves) or traverseWide (HBM-resident scenes: the 8-ary quantised nodes, wideNodePend, the
# per-lane LDS stack) -- the Src objects k_render_paths builds -- and every ray is compared with the oracle: a frame test says THAT a
# ray went wrong, this one says WHICH.

def _production_rays(data, osc, n, seed):
    """n camera rays, n secondary-like rays from their hit points (origins ON surfaces, random directions: the rays a path's bounces
    and shadow tests cast), n/8 nearly axis-parallel rays (|1/d| up to 1e9: inside the range the wide nodes take, b = (origin - o)/d
    huge), and -- LAST, so that only the final waves of the launch fall back to the binary records by the wave-wide vote -- 128 exactly
    axis-parallel rays."""
    cam = data.camera_desc(512, 384, 2, 2, 4)
    row, col, smp = _grid_samples(512, 384, 4, n, seed)
    o, d = ol.camera_rays(cam, row, col, smp)
    h = osc.closestHit(o, d)
    rs = np.random.RandomState(seed + 1)
    dirs = rs.normal(size=(n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True).astype(np.float32)
    o2 = np.where(h["hit"][:, None] > 0, h["point"] + np.float32(1e-3) * dirs, o).astype(np.float32)
    m = n // 8
    near = rs.normal(size=(m, 3)).astype(np.float32)
    ax = rs.randint(0, 3, m)
    near *= np.float32(1e-9) * (10.0 ** rs.uniform(0, 6, (m, 1))).astype(np.float32)
    near[np.arange(m), ax] = np.where(rs.rand(m) < 0.5, 1.0, -1.0)
    o3 = o2[rs.randint(0, n, m)]
    axis = np.zeros((128, 3), np.float32)
    axis[np.arange(128), np.arange(128) % 3] = np.where(np.arange(128) % 2, 1.0, -1.0)
    o4 = o2[rs.randint(0, n, 128)]
    return np.concatenate([o, o2, o3, o4]), np.concatenate([d, dirs, near, axis])


def _check_closest(sc, osc, o, d, traversal, want_source, what):
    g = sc.closestHit(o, d, traversal=traversal)
    assert sc.SOURCE_NAMES[sc.last_source] == want_source, f"{what}: ran {sc.SOURCE_NAMES[sc.last_source]}"
    r = osc.closestHit(o, d)
    bad = np.flatnonzero((g["hit"] != r["hit"]) | (g["prim"] != r["prim"]))
    assert len(bad) == 0, f"{what}: {len(bad)} rays hit another primitive; first ray {bad[0]}: o={o[bad[0]]} d={d[bad[0]]} gpu prim {g['prim'][bad[0]]} oracle {r['prim'][bad[0]]}"
    for k in ("t", "b1", "b2", "point", "normal", "uv"):
        assert_same_f32(g[k], r[k], f"{what}: {k}")
    return g


def _check_any(sc, osc, o, d, tmax, traversal, want_source, what):
    g = sc.anyHit(o, d, 0.0, tmax, traversal=traversal)
    assert sc.SOURCE_NAMES[sc.last_source] == want_source, f"{what}: ran {sc.SOURCE_NAMES[sc.last_source]}"
    r = osc.anyHit(o, d, 0.0, tmax)
    bad = np.flatnonzero(g != r)
    assert len(bad) == 0, f"{what}: {len(bad)} rays differ; first ray {bad[0]}: o={o[bad[0]]} d={d[bad[0]]} tmax={tmax[bad[0]]} gpu {g[bad[0]]} oracle {r[bad[0]]}"
    return g


@pytest.mark.parametrize("which", ["cornell", "mixed", "atrium_full"])
def test_per_ray_parity_through_the_shipped_traversals(which, request, gpu):
    """>= 100 k rays per scene and mode, closestHit AND anyHit: hit / prim / t / b1 / b2 / point / normal / uv bitwise equal to the oracle
    through the binary records (mode 0) and through the scene's production structure (mode 1): Cornell = the flat leaf list, the mixed
    scene and the 262 k-triangle atrium = the 8-ary quantised nodes."""
    if which == "atrium_full":
        data, sc, osc = request.getfixturevalue("atrium_full")
    else:
        data, sc, osc = request.getfixturevalue(which + "_pair")
    want = {"cornell": "leaf", "mixed": "wide", "atrium_full": "wide"}[which]
    info = sc.info()
    assert (want == "leaf") == bool(info["lds_resident"]) and (want != "wide" or info["wide_depth"] >= 2)
    o, d = _production_rays(data, osc, 48000, 21)
    assert len(o) >= 100000
    rs = np.random.RandomState(5)
    diag = float(np.linalg.norm(np.asarray(sc.bounds()[1]) - np.asarray(sc.bounds()[0])))
    tmax = rs.uniform(0.002 * diag, 0.8 * diag, len(o)).astype(np.float32)
    for traversal, src in ((sc.TRAVERSAL_BINARY, "binary"), (sc.TRAVERSAL_PRODUCTION, want)):
        g = _check_closest(sc, osc, o, d, traversal, src, f"{which} closestHit via {src}")
        assert g["hit"].mean() > 0.5
        a = _check_any(sc, osc, o, d, tmax, traversal, src, f"{which} anyHit via {src}")
        assert 0.05 < a.mean() < 0.95
    # finite intervals on closestHit too (the leaf list's phase A takes the open-ended form only for t.max = +inf)
    for traversal, src in ((sc.TRAVERSAL_PRODUCTION, want),):
        g = sc.closestHit(o[:20000], d[:20000], 0.001, 0.3 * diag, traversal=traversal)
        r = osc.closestHit(o[:20000], d[:20000], 0.001, 0.3 * diag)
        assert (g["hit"] == r["hit"]).all() and (g["prim"] == r["prim"]).all()
        assert_same_f32(g["t"], r["t"], f"{which} closestHit, finite interval, via {src}")


def test_per_ray_parity_on_equal_t_ties_and_lds_copies(gpu, cornell_pair):
    """(i) the equal-t tie scene (rays through grid vertices and along grid edges: several triangles at exactly the same distance, the
    first found in the reference's visiting order wins) through the 8-ary nodes, one- and four-primitive leaves; (ii) the LDS copy of
    the threaded records (source 1: what LDS-resident scenes with more than 32 leaves walk, and every LDS-resident scene under
    JTX_LEAF_WALK=0) on the Cornell box, named explicitly; (iii) asking for a structure the scene does not carry is an error."""
    for max_prims in (1, 4):
        data = _tie_scene(gpu)
        data.max_prims_in_node = max_prims
        sc = gpu.Scene(data); sc.buildBVH(max_prims)
        osc = ol.OracleScene(data)
        rs = np.random.RandomState(3)
        n = 60000
        gx = -4 + 8 * rs.randint(0, 25, n) / 24.0
        gy = 6 * rs.randint(0, 25, n) / 24.0
        target = np.stack([gx, gy, np.full(n, -3.0)], 1).astype(np.float32)
        o = np.tile(np.array([[0.0, 3.0, 9.0]], np.float32), (n, 1))
        o[: n // 3] += rs.uniform(-2, 2, (n // 3, 3)).astype(np.float32)         # a third from scattered origins (all regular rays)
        d = (target - o).astype(np.float32)
        g = _check_closest(sc, osc, o, d, sc.TRAVERSAL_PRODUCTION, "wide", f"ties, {max_prims} per leaf")
        assert g["hit"].mean() > 0.9
        tmax = rs.uniform(0.5, 1.5, n).astype(np.float32)                        # d is un-normalised: t = 1 is the grid plane
        _check_any(sc, osc, o, d, tmax, sc.TRAVERSAL_PRODUCTION, "wide", f"ties anyHit, {max_prims} per leaf")
        sc.destroy()
    data, sc, osc = cornell_pair
    o, d = _production_rays(data, osc, 16000, 33)
    _check_closest(sc, osc, o, d, sc.TRAVERSAL_SOURCE + 1, "lds", "cornell closestHit via the LDS copy")
    _check_any(sc, osc, o, d, np.full(len(o), 400.0, np.float32), sc.TRAVERSAL_SOURCE + 1, "lds", "cornell anyHit via the LDS copy")
    with pytest.raises(gpu.JtxMiError):
        sc.closestHit(o[:4], d[:4], traversal=sc.TRAVERSAL_SOURCE + 2)           # Cornell has no 8-ary nodes


def test_per_ray_parity_of_rays_that_miss_the_root_box(gpu, cornell_pair):
    """The leaf list asks the ROOT's box of a wave that holds nothing but camera rays and returns at once when every ray misses it
    (Scene::closestHit's first node test, scene.cpp:20-24); the per-ray entry point takes the same test.  Rays from the camera's side
    of the Cornell box: whole waves (64 consecutive rays) that miss the box, waves that all hit, waves that mix both, rays that graze
    the root's faces and edges -- hit / prim / t / b1 / b2 / point / normal / uv bitwise the oracle's, via the leaf list and the binary records."""
    data, sc, osc = cornell_pair
    lo, hi = np.asarray(sc.bounds()[0], np.float64), np.asarray(sc.bounds()[1], np.float64)
    ctr, ext = (lo + hi) / 2, (hi - lo)
    rs = np.random.RandomState(91)
    n = 64 * 600
    eye = ctr + np.array([0.0, 0.0, 2.5]) * ext.max()
    o = np.tile(eye.astype(np.float32), (n, 1))
    tgt = np.empty((n, 3))
    blk = np.arange(n) // 64
    kind = blk % 4                                    # per wave: 0 all miss (beside the box), 1 all hit, 2 mixed, 3 grazing the faces / edges
    away = ctr + np.stack([(1.5 + rs.uniform(0, 2, n)) * ext[0] * np.where(rs.rand(n) < 0.5, -1, 1), rs.uniform(-2, 2, n) * ext[1], rs.uniform(-1, 1, n) * ext[2]], 1)
    inside = lo + rs.uniform(0.05, 0.95, (n, 3)) * ext
    mixed = ctr + rs.uniform(-1.2, 1.2, (n, 3)) * ext
    graze = inside.copy()
    ax = rs.randint(0, 2, n)
    graze[np.arange(n), ax] = np.where(rs.rand(n) < 0.5, lo[ax], hi[ax]) + rs.uniform(-1e-4, 1e-4, n) * ext[ax]
    graze[:, 2] = hi[2]
    for k, t in enumerate((away, inside, mixed, graze)):
        tgt[kind == k] = t[kind == k]
    d = (tgt - o).astype(np.float32)
    g = _check_closest(sc, osc, o, d, sc.TRAVERSAL_PRODUCTION, "leaf", "rays from outside the root box")
    hit = g["hit"].reshape(-1, 64)
    assert (hit[0::4].sum(1) == 0).all() and (hit[1::4].sum(1) == 64).all(), "waves of all-missing / all-hitting rays"
    m = hit[2::4].sum(1)
    assert ((m > 0) & (m < 64)).any()
    _check_closest(sc, osc, o, d, sc.TRAVERSAL_BINARY, "binary", "rays from outside the root box")
    # from scattered origins too (no common eye), and with un-normalised directions scaled per ray
    o2 = (eye + rs.uniform(-1, 1, (n, 3)) * ext * 0.5).astype(np.float32)
    d2 = ((tgt - o2) * rs.uniform(0.25, 4.0, (n, 1))).astype(np.float32)
    _check_closest(sc, osc, o2, d2, sc.TRAVERSAL_PRODUCTION, "leaf", "scattered origins outside the root box")
    # origins beyond the leaf list's range (|o| > 2^60: plane - o must stay finite for the fma form of its box test): those waves walk the
    # binary records -- same answers; and a list is not built at all for a scene whose planes lie out there
    o3 = o2.copy(); o3[::3] *= np.float32(3e25)
    d3 = (tgt - o3.astype(np.float64)).astype(np.float32)
    _check_closest(sc, osc, o3, d3, sc.TRAVERSAL_PRODUCTION, "leaf", "origins beyond 2^60")
    _check_any(sc, osc, o3, d3, np.full(n, 0.999, np.float32), sc.TRAVERSAL_PRODUCTION, "leaf", "origins beyond 2^60, anyHit")
    far = gpu.scenes.cornell()
    m = np.eye(4, dtype=np.float32); m[0, 3] = 4e18
    for mesh in far.meshes:
        mesh["transform"] = np.ascontiguousarray(m @ np.asarray(mesh["transform"], np.float32), np.float32)
    fsc = gpu.Scene(far); fsc.buildBVH()
    assert fsc.info()["lds_resident"] and fsc.closestHit(o[:64], d[:64], traversal=fsc.TRAVERSAL_PRODUCTION) is not None
    assert fsc.SOURCE_NAMES[fsc.last_source] == "lds", "a scene beyond the leaf list's range walks the LDS copy of the binary records"
    fsc.destroy()


def test_soak_of_the_create_render_rebuild_destroy_path(gpu):
    """tools/soak.py for a minute (the full 300 iterations are recorded in profiles/r05_soak.txt): scenes created,
    rendered into page-locked film buffers, edited, rebuilt on the device, rendered again and destroyed, by turns and at changing sizes --
    every frame bit-identical to the first of its kind, the abort log empty.  In a child process: a GPU fault there is this test's
    failure, not the session's end."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    log = os.path.join(root, "gpurun_out", "jtx_abort_soak_test.log")
    if os.path.exists(log):
        os.remove(log)
    env = dict(os.environ, JTX_ABORT_LOG=log)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), "60", "45", "--no-rehearsal"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "0 faults, abort log empty" in r.stdout and "soak: " in r.stdout
    n = int(r.stdout.split("soak: ")[1].split()[0])
    assert n >= 20


def test_rebuild_spare_set_is_accounted_and_can_be_released(gpu):
    """the second set of structures a device rebuild writes (about the geometry's own device memory again, plus the
    builder's scratch) shows in scene_info.rebuild_spare_bytes, jtx_mi_scene_release_rebuild gives it back, and the next rebuild
    allocates it anew -- same tree, same frame.  wide_bytes is what the kernels walk (jtx_mi_scene_get_wide agrees)."""
    data = gpu.scenes.atrium(target_tris=12000)
    sc = gpu.Scene(data); sc.buildBVH()
    i0 = sc.info()
    assert i0["rebuild_spare_bytes"] == 0 and i0["wide_bytes64"] == i0["wide_bytes"] == 16 * len(sc.wide()) > 0
    sc.reserveRebuild()
    i1 = sc.info()
    assert i1["rebuild_spare_bytes"] > 0.5 * i0["device_bytes"] and i1["device_bytes"] == i0["device_bytes"]
    m = np.eye(4, dtype=np.float32); m[1, 3] = 1.5
    sc.setTransform(0, m)
    sc.rebuildBVHOnDevice()
    n1, r1 = sc.bvh()
    ref = gpu.StaticCamera(160, 90, data.camera, 2, 2, 5); ref.render(sc, count_rays=False)
    sc.releaseRebuild()
    assert sc.info()["rebuild_spare_bytes"] == 0
    g = gpu.StaticCamera(160, 90, data.camera, 2, 2, 5); g.render(sc, count_rays=False)
    assert_same_f32(g.acc_, ref.acc_, "frame after the spare set was released")
    sc.rebuildBVHOnDevice()                                           # allocates the set again; the geometry did not change: the same tree
    n2, r2 = sc.bvh()
    _same_tree(n1, r1, n2, r2, "rebuild after release")
    assert sc.info()["rebuild_spare_bytes"] > 0
    g.render(sc, count_rays=False)
    assert_same_f32(g.acc_, ref.acc_, "frame after the rebuild that followed the release")
    sc.destroy()


def test_cancellation_with_passes_in_flight_leaves_a_prefix_of_the_strata(gpu, mixed_pair):
    """jtx_mi_render traces all passes of a frame in ONE launch (round 6; three passes in flight in round 5) with a resolver kernel beside
    it; a cancellation from another thread may catch it anywhere.  Whatever it catches: the film must hold EXACTLY the strata
    [0, currentSample_) -- passes enter the film in order or not at all (k_resolve_progressive: a pass counts when all its chunks were
    dealt and no wave holds a path of it) -- and the RGB8 preview must be that film's.  Twelve cancellations at different moments of a
    24-pass render, one stratum per pass, plus one from the callback."""
    import threading, time
    data, sc, osc = mixed_pair
    W, H = 1280, 720                                                          # (one launch for all 24 passes takes ~10 ms at this size: room for the timers)
    refs = {}

    def ref(n):
        if n not in refs:
            c = gpu.StaticCamera(W, H, data.camera, 6, 4, 6)
            if n:
                c.render(sc, sample_begin=0, sample_end=n)
            refs[n] = (c.acc_.copy(), c.img_.copy())
        return refs[n]
    full = gpu.StaticCamera(W, H, data.camera, 6, 4, 6); full.samplesPerPass_ = 1
    full.render(sc, progress=lambda c, t: None)                               # (page-locks its film, loads the kernels)
    t0 = time.perf_counter(); full.render(sc, progress=lambda c, t: None); t_full = time.perf_counter() - t0
    assert full.currentSample_ == 24
    assert_same_f32(full.acc_, ref(24)[0], "24 passes, one launch")
    seen = set()
    cams = []                                                                 # (kept alive: releasing a camera's page-locked film -- hipHostUnregister --
    for k in range(12):                                                       #  right before the next render delays that render's start by milliseconds)
        cam = gpu.StaticCamera(W, H, data.camera, 6, 4, 6); cam.samplesPerPass_ = 1
        cams.append(cam)
        cam._pin()                                                            # (before the clock runs)
        timer = threading.Timer(t_full * (0.05 + 0.08 * k), cam.terminateRender)
        timer.start(); cam.render(sc, progress=lambda c, t: None); timer.join()
        n = cam.currentSample_
        seen.add(n)
        acc, img = ref(n)
        assert_same_f32(cam.acc_, acc, f"film after a cancellation that left {n} strata")
        if n:
            assert (cam.img_ == img).all(), f"preview after a cancellation that left {n} strata"
    assert len(seen) >= 3 and min(seen) < 24, seen                            # the cancellations did land at different passes
    cam = gpu.StaticCamera(W, H, data.camera, 6, 4, 6); cam.samplesPerPass_ = 1
    cam.render(sc, progress=lambda c, t: cam.terminateRender() if c == 5 else None)
    n = cam.currentSample_
    assert 5 <= n <= 24                                                       # (the launch does not wait for the callback: passes behind the 5th may be in)
    assert_same_f32(cam.acc_, ref(n)[0], f"film after a callback cancel at 5 that left {n} strata")


def test_progressive_launch_of_a_tile_shard_with_previews(gpu, cornell_pair):
    """a tile-sharded progressive render (tile_world = 3, one stratum per pass, a callback) must deliver exact zeros in the tiles
    it does not own -- in every preview and in the final image -- and its own tiles bit for bit (round 5's three preview buffers came
    uninitialised from hipMalloc; round 6 renders all passes in one launch into ONE preview buffer, cleared per call).  Also: an odd pass
    size that does not divide the strata, and the per-pass loop (JTX_PROGRESSIVE_LAUNCH=0 semantics are covered by the counted renders)."""
    data, sc, osc = cornell_pair
    W, H = 200, 120
    whole = gpu.StaticCamera(W, H, data.camera, 4, 3, 4); whole.render(sc)
    owner = (np.arange(H)[:, None] // 32) * ((W + 31) // 32) + np.arange(W)[None, :] // 32
    for rank in range(3):
        mine = (owner % 3) == rank
        cam = gpu.StaticCamera(W, H, data.camera, 4, 3, 4); cam.samplesPerPass_ = 1
        previews = []
        cam.render(sc, tile_rank=rank, tile_world=3, progress=lambda c, t: previews.append((c, cam.img_.copy())))
        assert [c for c, _ in previews] == list(range(1, 13))
        for c, img in previews:
            assert not img[~mine].any(), f"preview {c} of shard {rank}: bytes in tiles it does not own"
        assert not cam.img_[~mine].any() and not cam.acc_[~mine].any()
        assert_same_f32(cam.acc_[mine], whole.acc_[mine], f"shard {rank} of 3, progressive")
        assert (cam.img_[mine] == whole.img_[mine]).all()
    # a shard WITHOUT tiles (more ranks than 32x32 tiles: 28 tiles here, rank 30 of 32): nothing to launch, every callback, an empty film
    none = gpu.StaticCamera(W, H, data.camera, 4, 3, 4); none.samplesPerPass_ = 4
    seen = []
    none.render(sc, tile_rank=30, tile_world=32, progress=lambda c, t: seen.append(c))
    assert seen == [4, 8, 12] and not none.acc_.any() and not none.img_.any()
    odd = gpu.StaticCamera(W, H, data.camera, 4, 3, 4); odd.samplesPerPass_ = 5
    seen = []
    odd.render(sc, progress=lambda c, t: seen.append(c))
    assert seen == [5, 10, 12]
    assert_same_f32(odd.acc_, whole.acc_, "passes of 5 strata, the last one shorter"); assert (odd.img_ == whole.img_).all()
    # a RESUMED progressive render (the film of strata [0, 5) uploaded, [5, 12) in passes of 2) and one whose records do not fit the cap
    # (1 MB holds two strata of this frame: four progressive launches, one after the other)
    res = gpu.StaticCamera(W, H, data.camera, 4, 3, 4); res.samplesPerPass_ = 2
    res.render(sc, sample_begin=0, sample_end=5)
    seen = []
    res.render(sc, sample_begin=5, sample_end=12, progress=lambda c, t: seen.append(c))
    assert seen == [7, 9, 11, 12]
    assert_same_f32(res.acc_, whole.acc_, "resumed progressive render"); assert (res.img_ == whole.img_).all()
    cap = gpu.StaticCamera(W, H, data.camera, 4, 3, 4); cap.samplesPerPass_ = 1
    seen = []
    cap.render(sc, progress=lambda c, t: seen.append(c), max_record_mb=1)
    assert seen == list(range(1, 13))
    assert_same_f32(cap.acc_, whole.acc_, "progressive render under a 1 MB record cap"); assert (cap.img_ == whole.img_).all()
    # very many passes: 190 x 190 strata at one per pass -- more than one launch's 15-bit pass counter holds, so two launches
    many = gpu.StaticCamera(32, 32, data.camera, 190, 190, 3); many.samplesPerPass_ = 1
    n = [0, 0]
    def tick(c, t):
        assert c == n[1] + 1
        n[0] += 1; n[1] = c
    many.render(sc, progress=tick)
    batch = gpu.StaticCamera(32, 32, data.camera, 190, 190, 3); batch.render(sc)
    assert n == [190 * 190, 190 * 190] and many.currentSample_ == 190 * 190
    assert_same_f32(many.acc_, batch.acc_, "36 100 passes of one stratum"); assert (many.img_ == batch.img_).all()
    # a caller without an RGB8 image (img_rgb = NULL, through the C-ABI directly): callbacks and the film as with one
    import ctypes as C
    capi = gpu._capi
    acc_only = np.zeros((H, W, 3), np.float32)
    o = capi.RenderOpts(); o.samples_per_tick = 5
    seen = []
    cb = capi.PROGRESS_CB(lambda cur, tot, _u: seen.append((cur, tot)) or 0)
    d = whole.desc()
    capi.check(capi.load().jtx_mi_render(sc.handle, C.byref(d), C.byref(o), acc_only.ctypes.data_as(C.POINTER(C.c_float)), None, cb, None))
    assert seen == [(5, 12), (10, 12), (12, 12)]
    assert_same_f32(acc_only, whole.acc_, "progressive render without an image buffer")
    # the smallest and a ragged frame (one pixel; 33 x 65: tiles that overhang on both sides)
    for (w, h) in ((1, 1), (33, 65)):
        a = gpu.StaticCamera(w, h, data.camera, 3, 3, 4); a.samplesPerPass_ = 2
        seen = []
        a.render(sc, progress=lambda c, t: seen.append(c))
        b = gpu.StaticCamera(w, h, data.camera, 3, 3, 4); b.render(sc)
        assert seen == [2, 4, 6, 8, 9]
        assert_same_f32(a.acc_, b.acc_, f"progressive {w}x{h} frame"); assert (a.img_ == b.img_).all()


_ONE_STREAM_CHILD = r"""
import numpy as np
import jtx_pathtracer_amd as gpu
gpu._capi.check(gpu._capi.load().jtx_mi_set_device(0))
data = gpu.scenes.cornell()
sc = gpu.Scene(data); sc.buildBVH()
ref = gpu.StaticCamera(320, 200, data.camera, 4, 4, 4); ref.render(sc)
for spp_pass in (1, 3, 8):
    cam = gpu.StaticCamera(320, 200, data.camera, 4, 4, 4); cam.samplesPerPass_ = spp_pass
    seen = []
    cam.render(sc, progress=lambda c, t: seen.append(c))
    assert seen == list(range(spp_pass, 16, spp_pass)) + [16], seen
    assert np.array_equal(cam.acc_.view(np.uint32), ref.acc_.view(np.uint32)) and np.array_equal(cam.img_, ref.img_)
sc.destroy()
print("one-stream: ok")
"""


def test_progressive_launch_when_the_two_kernels_cannot_run_side_by_side(gpu, tmp_path):
    """The progressive launch is two kernels on two streams, and HIP does not promise that streams run side by side (they may share a
    hardware queue: tools/soak.py met that mapping in round 6 -- with the resolver launched FIRST the path kernel queued behind it and the
    resolver waited for it).  The path kernel is launched first and waits for nobody, so the serialised case is merely late with its
    previews: here both kernels are put on ONE stream (a hook only libjtx_mi_testhooks.so carries) -- same film bit for bit, every
    callback delivered, no wait."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "jtx-pathtracer_amd", "libjtx_mi_testhooks.so")
    assert os.path.exists(lib), "build it with __graft_entry__.build() (jtx.build_test_hooks)"
    script = tmp_path / "one_stream_child.py"
    script.write_text(_ONE_STREAM_CHILD)
    env = dict(os.environ, JTX_MI_LIB=lib, JTX_TEST_PROGRESSIVE_ONE_STREAM="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "one-stream: ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    assert time.perf_counter() - t0 < 50.0                                      # (the resolver's bounded wait is a minute: it was never needed)


_NO_PATH_KERNEL_CHILD = r"""
import os, time
import numpy as np
import jtx_pathtracer_amd as gpu
gpu._capi.check(gpu._capi.load().jtx_mi_set_device(0))
data = gpu.scenes.cornell()
sc = gpu.Scene(data); sc.buildBVH()
ref = gpu.StaticCamera(320, 200, data.camera, 4, 4, 4); ref.samplesPerPass_ = 2
ref.render(sc, progress=lambda c, t: None)
os.environ["JTX_TEST_PROGRESSIVE_NO_PATH_KERNEL"] = "1"          # (read at every launch)
os.environ["JTX_TEST_RESOLVER_PATIENCE_MS"] = "300"
cam = gpu.StaticCamera(320, 200, data.camera, 4, 4, 4); cam.samplesPerPass_ = 2
seen = []
t0 = time.perf_counter()
try:
    cam.render(sc, progress=lambda c, t: seen.append(c))
    raise SystemExit("the render without a path kernel returned success")
except RuntimeError as e:
    dt = time.perf_counter() - t0
    assert "gave up" in str(e), str(e)
assert seen == [] and 0.25 < dt < 10.0, (seen, dt)               # no pass was ever reported; the wait was the patience, not a minute
del os.environ["JTX_TEST_PROGRESSIVE_NO_PATH_KERNEL"]
# ... but a path kernel that is merely LATE -- its stream held for 0.5 s, as behind another process' long launch, five times the resolver's patience here --
# is waited for: the host vouches for it as long as it has not ended
os.environ["JTX_TEST_RESOLVER_PATIENCE_MS"] = "100"
os.environ["JTX_TEST_PROGRESSIVE_DELAY_PATH_MS"] = "500"
late = gpu.StaticCamera(320, 200, data.camera, 4, 4, 4); late.samplesPerPass_ = 2
seen = []
t0 = time.perf_counter()
late.render(sc, progress=lambda c, t: seen.append(c))
assert time.perf_counter() - t0 > 0.45 and seen == list(range(2, 17, 2)), (time.perf_counter() - t0, seen)
assert np.array_equal(late.acc_.view(np.uint32), ref.acc_.view(np.uint32)) and np.array_equal(late.img_, ref.img_)
del os.environ["JTX_TEST_PROGRESSIVE_DELAY_PATH_MS"]
again = gpu.StaticCamera(320, 200, data.camera, 4, 4, 4); again.samplesPerPass_ = 2
seen = []
again.render(sc, progress=lambda c, t: seen.append(c))
assert seen == list(range(2, 17, 2)), seen
assert np.array_equal(again.acc_.view(np.uint32), ref.acc_.view(np.uint32)) and np.array_equal(again.img_, ref.img_)
sc.destroy()
print("no-path-kernel: ok %.2f s" % dt)
"""


def test_progressive_resolver_gives_up_when_the_path_kernel_never_comes(gpu, tmp_path):
    """The resolver of a progressive launch waits for a kernel it does not control; the wait is bounded (a minute without a chunk
    fetched or a wave's word moving), it then ends with the passes it has and says so, and jtx_mi_render returns an error.  Exercised
    with the hooks library: the path launch is skipped and the patience shortened to 0.3 s -- the error comes, no pass is reported,
    nothing hangs, and the scene renders the right film afterwards.  A path kernel that is merely late (its stream held for five times the
    patience) is waited for: the host's polling loop vouches for it in host-mapped memory until it has ended."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "jtx-pathtracer_amd", "libjtx_mi_testhooks.so")
    assert os.path.exists(lib), "build it with __graft_entry__.build() (jtx.build_test_hooks)"
    script = tmp_path / "no_path_kernel_child.py"
    script.write_text(_NO_PATH_KERNEL_CHILD)
    env = dict(os.environ, JTX_MI_LIB=lib, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=200, env=env)
    assert r.returncode == 0 and "no-path-kernel: ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_two_scenes_render_progressively_at_the_same_time(gpu):
    """Different scenes are independent (jtx_mi.h): two host threads, each with a scene and a camera of its own, render progressive frames
    -- a callback after every stratum -- at the same time on one device.  Each launch wants the whole chip (persistent waves) and brings a
    resolver: whichever path kernel comes second waits its turn, its resolver beside it does not give up on it, and every film is the
    film of the scene rendered alone."""
    import threading
    jobs = []
    for make, (w, h), (xs, ys) in ((gpu.scenes.cornell, (640, 360), (4, 4)), (lambda: gpu.scenes.atrium(target_tris=20000), (480, 272), (3, 3))):
        data = make()
        sc = gpu.Scene(data); sc.buildBVH()
        ref = gpu.StaticCamera(w, h, data.camera, xs, ys, 5); ref.render(sc)
        jobs.append((data, sc, ref, w, h, xs, ys))
    errors = []
    def work(job):
        data, sc, ref, w, h, xs, ys = job
        try:
            for _ in range(6):
                cam = gpu.StaticCamera(w, h, data.camera, xs, ys, 5); cam.samplesPerPass_ = 1
                seen = []
                cam.render(sc, progress=lambda c, t: seen.append(c))
                assert seen == list(range(1, xs * ys + 1)), seen
                assert np.array_equal(cam.acc_.view(np.uint32), ref.acc_.view(np.uint32)) and np.array_equal(cam.img_, ref.img_)
                cam._unpin()
        except BaseException as e:          # noqa: BLE001 (reported by the main thread)
            errors.append(repr(e))
    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors
    for j in jobs: j[1].destroy()


def test_frame_slot_memory_is_accounted_and_can_be_released(gpu, cornell_pair):
    """VERDICT r5 missing 5: the frame slots' working memory (the per-path radiance records of the persistent path kernel) is reported
    (scene_info.frame_slot_bytes, not part of device_bytes), capped per launch by opts.max_record_mb, and given back by
    jtx_mi_scene_release_frames; the next render allocates it again and gives the same film."""
    data, _, osc = cornell_pair
    sc = gpu.Scene(data); sc.buildBVH()
    assert sc.info()["frame_slot_bytes"] == 0
    cam = gpu.StaticCamera(320, 200, data.camera, 4, 4, 4); cam.render(sc)
    held = sc.info()["frame_slot_bytes"]
    tiles = ((320 + 31) // 32) * ((200 + 31) // 32)
    assert held >= tiles * 1024 * 16 * 16                                    # 16 strata x the owned pixel slots x 16 B
    ref = cam.acc_.copy()
    sc.releaseFrames()
    assert sc.info()["frame_slot_bytes"] == 0
    cam.render(sc)
    assert_same_f32(cam.acc_, ref, "frame after release_frames")
    assert sc.info()["frame_slot_bytes"] >= tiles * 1024 * 16 * 16
    # the cap as a field of the options: 1 MB holds 0.9 strata of this frame -> one stratum per launch, the same film
    sc.releaseFrames()
    capped = gpu.StaticCamera(320, 200, data.camera, 4, 4, 4); capped.render(sc, max_record_mb=2)
    assert_same_f32(capped.acc_, ref, "frame under a 2 MB record cap")
    assert sc.info()["frame_slot_bytes"] <= (2 << 20) + 65536
    sc.destroy()


def test_headline_frame_against_the_oracle_in_every_pixel(gpu, cornell_pair):
    """The frame `bench.py` times (C2: Cornell 1920x1080, 64 spp, depth 8) against a WHOLE-FRAME run of the CPU oracle -- all 2 073 600
    pixels, film words, RGB8 bytes and every ray counter (0.73 G rays on the host's cores: ~20 s) -- where the other full-size tests
    compare the whole frame with the counting kernel and 256 pixels with the oracle.  Through the pipelined loop
    the bench runs: three frames in flight, the middle one checked."""
    import time
    import torch
    data, sc, osc = cornell_pair
    W, H = 1920, 1080
    cam = data.camera_desc(W, H, 8, 8, 8)
    t0 = time.perf_counter()
    acc, img, cnt = osc.render(cam, threads=min(os.cpu_count() or 1, 64))
    t_oracle = time.perf_counter() - t0
    assert cnt["n_camera"] == W * H * 64 and cnt["n_closest"] + cnt["n_any"] == 727984390           # 5.49 rays per sample
    dev = torch.device("cuda", 0)
    pipe = gpu.distributed.ShardPipeline(sc, cam, 0, 1, dev, None, integrator=1)
    for k in range(3):
        pipe.step(last=(k == 2))
    torch.cuda.synchronize()
    a = pipe.accs[1].cpu().numpy().reshape(H, W, 3); i = pipe.imgs[1].cpu().numpy().reshape(H, W, 3)
    assert np.array_equal(a.view(np.uint32), acc.view(np.uint32)), f"{int((a.view(np.uint32) != acc.view(np.uint32)).any(-1).sum())} pixels differ"
    assert np.array_equal(i, img)
    # ... and the counting kernel's tallies (the bench line's numerators) are the oracle's, the per-class ones included
    acc_c, img_c, cnt_g = _frame_on_device(gpu, sc, cam, 1, count=True)
    assert cnt_g == cnt and np.array_equal(acc_c.view(np.uint32), acc.view(np.uint32))
    # ... and the frame as the reference's UI renders it -- a callback after every pass of ONE stratum (samplesPerPass_ = 1, camera.hpp:181):
    # one progressive launch (k_render_paths<PROG> beside k_resolve_progressive), the same film in every pixel, 64 callbacks in order
    ui = gpu.StaticCamera(W, H, data.camera, 8, 8, 8); ui.samplesPerPass_ = 1
    seen = []
    ui.render(sc, progress=lambda c, t: seen.append(c))
    assert seen == list(range(1, 65))
    assert np.array_equal(ui.acc_.view(np.uint32), acc.view(np.uint32)) and np.array_equal(ui.img_, img)
    assert t_oracle < 240


@pytest.mark.parametrize("which", ["mixed", "atrium_full"])
def test_timed_workloads_against_the_oracle_in_every_pixel(gpu, which, request):
    """... and the other two timed workloads: C5 (mixed materials, 1920x1080, all 128 strata: ~30 s of oracle on the host's cores) and C3
    (the 262 k-triangle atrium at 1920x1080: strata [0, 16) of its 64, ~20 s -- the whole frame would take the oracle 80 s) through the
    uncounted kernels of the 8-ary BVH: every pixel's film words and RGB8 bytes, and through the counting kernel every ray counter."""
    if which == "atrium_full":
        data, sc, osc = request.getfixturevalue("atrium_full")
        dims, s_end = (1920, 1080, 8, 8, 8), 16
    else:
        data = gpu.scenes.mixed()
        sc = gpu.Scene(data); sc.buildBVH()
        osc = ol.OracleScene(data)
        dims, s_end = (1920, 1080, 16, 8, 8), 128
    W, H = dims[0], dims[1]
    cam = data.camera_desc(*dims)
    acc, img, cnt = osc.render(cam, threads=min(os.cpu_count() or 1, 64), sample_begin=0, sample_end=s_end)
    g = gpu.StaticCamera(W, H, data.camera, dims[2], dims[3], dims[4])
    g.render(sc, count_rays=False, sample_begin=0, sample_end=s_end)
    assert sc.info()["wide_depth"] >= 2
    diff = (np.asarray(g.acc_).view(np.uint32) != acc.view(np.uint32)).any(-1)
    assert not diff.any(), f"{int(diff.sum())} of {W * H} pixels differ"
    assert np.array_equal(g.img_, img)
    g.render(sc, count_rays=True, sample_begin=0, sample_end=s_end)
    assert g.counters == cnt
    if which != "atrium_full":
        sc.destroy()
