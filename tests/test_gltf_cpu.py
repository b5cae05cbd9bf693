"""glTF / GLB ingestion: PNG codec, stbi_loadf float conversion, loader.cpp semantics, round trip
through this package's own writer, and structural checks on the reference's helmet.glb where it is present."""
import os
import struct
import zlib

import numpy as np
import pytest

import jtx_pathtracer_amd as jtx
from jtx_pathtracer_amd import gltf, scenes

HELMET = "/root/reference/src/assets/scenes/helmet.glb"


def _png_with_filters(img):
    """encode with scanline filter (y % 5): exercises none / sub / up / average / paeth in the decoder"""
    h, w, c = img.shape
    raw = bytearray()
    prev = np.zeros(w * c, np.int32)
    for y in range(h):
        line = img[y].reshape(-1).astype(np.int32)
        ft = y % 5
        out = np.zeros_like(line)
        for x in range(len(line)):
            a = line[x - c] if x >= c else 0
            b = prev[x]
            cc = prev[x - c] if x >= c else 0
            if ft == 0:
                pr = 0
            elif ft == 1:
                pr = a
            elif ft == 2:
                pr = b
            elif ft == 3:
                pr = (a + b) >> 1
            else:
                pa, pb, pc = abs(b - cc), abs(a - cc), abs(a + b - 2 * cc)
                pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
            out[x] = (line[x] - pr) & 255
        raw += bytes([ft]) + out.astype(np.uint8).tobytes()
        prev = line
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xffffffff)

    half = len(raw) // 2
    comp = zlib.compress(bytes(raw))
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + \
        chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b"")


@pytest.mark.parametrize("c", [1, 2, 3, 4])
def test_png_codec_round_trip_and_filters(c):
    rs = np.random.RandomState(c)
    img = rs.randint(0, 256, (13, 9, c)).astype(np.uint8)
    assert np.array_equal(gltf.decode_png(gltf.encode_png(img)), img)
    assert np.array_equal(gltf.decode_png(_png_with_filters(img)), img)


def test_stbi_loadf_conversion():
    px = np.array([[[0, 128, 255, 64]]], np.uint8)
    f = gltf.ldr_to_float(px)
    assert f.dtype == np.float32 and f.shape == (1, 1, 4)
    assert f[0, 0, 0] == 0.0 and f[0, 0, 2] == 1.0
    assert abs(f[0, 0, 1] - (128 / 255.0) ** 2.2) < 1e-7
    assert f[0, 0, 3] == np.float32(64) / np.float32(255)          # alpha stays linear
    g = gltf.ldr_to_float(np.array([[[10, 200, 30]]], np.uint8))
    assert np.allclose(g[0, 0], (np.array([10, 200, 30]) / 255.0) ** 2.2, atol=1e-7)


def _rot_x(deg):
    a = np.deg2rad(deg)
    m = np.eye(4)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
    return m


def _source_scene():
    s = scenes.SceneData("src")
    rs = np.random.RandomState(0)
    tex = rs.randint(0, 256, (8, 16, 3)).astype(np.uint8)
    mr = rs.randint(0, 256, (4, 4, 4)).astype(np.uint8)
    s.materials = [scenes.material(scenes.METALLIC_ROUGHNESS, (1, 1, 1), alpha_x=0.25, alpha_y=0.6, albedo_tex=0, mr_tex=1),
                   scenes.material(scenes.METALLIC_ROUGHNESS, (1, 1, 1), alpha_x=1.0, alpha_y=0.1)]
    g = scenes._grid_quad((-2, 0, -2), (4, 0, 0), (0, 0, 4), 3, 2, (0, 1, 0))
    s.add_mesh(g[0], g[1], g[2], 0, uvs=g[3], name="floor")
    c = scenes._column(0.0, 0.0, 0.0, 2.0, 0.5, 8, 3)
    s.add_mesh(c[0], c[1], c[2], 1, uvs=c[3], name="column")
    return s, [tex, mr]


def test_glb_round_trip_with_node_transforms(tmp_path):
    src, tex = _source_scene()
    mats = [None, _rot_x(90.0) @ np.diag([2.0, 2.0, 2.0, 1.0])]
    mats[1][:3, 3] = (1.0, 0.5, -0.25)
    path = str(tmp_path / "scene.glb")
    gltf.write_glb(path, src, textures_u8=tex, node_matrices=mats)
    got = gltf.load_gltf(path, background=(0.1, 0.2, 0.3))
    assert got.sky == (0.1, 0.2, 0.3) and got.camera["center"] == (0, 0, 8) and got.camera["yfov"] == 20.0   # scene.cpp:181-188
    assert len(got.meshes) == 2 and got.num_triangles == src.num_triangles
    assert [m["name"] for m in got.meshes] == ["floor", "column"]
    # mesh 0: identity node
    assert np.array_equal(got.meshes[0]["vertices"], src.meshes[0]["vertices"])
    assert np.array_equal(got.meshes[0]["uvs"], src.meshes[0]["uvs"])          # FlipUVs undone by the writer's 1 - v
    assert np.array_equal(got.meshes[0]["indices"], src.meshes[0]["indices"])
    # mesh 1: PreTransformVertices -- points by the matrix, normals by its inverse transpose, re-normalised
    m = mats[1]
    want_p = (src.meshes[1]["vertices"].astype(np.float64) @ m[:3, :3].T + m[:3, 3]).astype(np.float32)
    assert np.array_equal(got.meshes[1]["vertices"], want_p)
    want_n = src.meshes[1]["normals"].astype(np.float64) @ np.linalg.inv(m[:3, :3])
    want_n /= np.linalg.norm(want_n, axis=1, keepdims=True)
    assert np.allclose(got.meshes[1]["normals"], want_n, atol=1e-6)
    assert np.allclose(np.linalg.norm(got.meshes[1]["normals"], axis=1), 1.0, atol=1e-6)
    # materials: METALLIC_ROUGHNESS, white, factors in the alpha fields (loader.cpp:130-138)
    for a, b in zip(got.materials, src.materials):
        assert a["type"] == scenes.METALLIC_ROUGHNESS and tuple(a["albedo"]) == (1.0, 1.0, 1.0)
        assert (a["alpha_x"], a["alpha_y"], a["albedo_tex"], a["mr_tex"]) == (b["alpha_x"], b["alpha_y"], b["albedo_tex"], b["mr_tex"])
    assert [m["material"] for m in got.meshes] == [0, 1]
    # textures: stbi_loadf floats of the embedded PNGs, file channel count kept
    assert len(got.textures) == 2
    assert np.array_equal(got.textures[0], gltf.ldr_to_float(tex[0])) and got.textures[1].shape == (4, 4, 4)
    # Triangle refs: one per face, mesh by mesh
    refs = got.tri_refs()
    assert (refs[: len(src.meshes[0]["indices"]), 1] == 0).all() and (refs[len(src.meshes[0]["indices"]):, 1] == 1).all()
    # and the result is a valid scene for the host BVH builder
    nodes, _, depth = jtx.api.bvh_build_host(got)
    assert len(nodes) > 10 and depth > 2


def test_gen_normals_and_default_material(tmp_path):
    """a primitive without NORMAL gets flat face normals on un-shared vertices (GenNormals); without a material the
    pink Lambert of loader.cpp:206-209"""
    import json
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], np.float32)
    idx = np.array([0, 1, 2, 2, 1, 3], np.uint16)
    blob = pos.tobytes() + idx.tobytes()
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0, "translation": [0, 0, 2]}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1}]}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"},
                         {"bufferView": 1, "componentType": 5123, "count": 6, "type": "SCALAR"}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 48}, {"buffer": 0, "byteOffset": 48, "byteLength": 12}],
           "buffers": [{"byteLength": len(blob), "uri": "geo.bin"}]}
    (tmp_path / "geo.bin").write_bytes(blob)
    (tmp_path / "s.gltf").write_text(json.dumps(doc))
    s = gltf.load_gltf(str(tmp_path / "s.gltf"))
    m = s.meshes[0]
    assert m["vertices"].shape == (6, 3) and np.array_equal(m["indices"], [[0, 1, 2], [3, 4, 5]])
    assert np.array_equal(m["vertices"][:, 2], np.full(6, 2.0, np.float32))
    assert np.array_equal(m["normals"], np.tile(np.array([[0, 0, 1]], np.float32), (6, 1)))
    assert m["uvs"] is None
    assert s.materials[m["material"]]["type"] == scenes.DIFFUSE and tuple(s.materials[m["material"]]["albedo"]) == (1.0, 0.3, 0.5)


@pytest.mark.skipif(not os.path.exists(HELMET), reason="reference assets are only present in the build container")
def test_reference_helmet_glb_structure():
    s = gltf.load_gltf(HELMET)                               # its five maps are JPEG (one progressive): decoded since round 2
    assert len(s.meshes) == 1 and s.num_triangles == 46356 // 3 and len(s.meshes[0]["vertices"]) == 14556
    m = s.materials[s.meshes[0]["material"]]
    assert m["type"] == scenes.METALLIC_ROUGHNESS and (m["alpha_x"], m["alpha_y"]) == (1.0, 1.0)
    assert m["albedo_tex"] == 0 and m["mr_tex"] == 1 and len(s.textures) == 2      # baseColor + metallicRoughness (loader.cpp:115-126)
    v, n = s.meshes[0]["vertices"], s.meshes[0]["normals"]
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-4)
    # the node rotates by 90 degrees about x: the helmet's long axis ends up along z... just check the bake happened
    ext = v.max(0) - v.min(0)
    assert ext.min() > 0.5 and ext.max() < 3.0
    uv = s.meshes[0]["uvs"]
    assert uv is not None and np.isfinite(uv).all() and uv.shape == (14556, 2)      # v wraps (the file stores v in [1, 2])
    nodes, _, depth = jtx.api.bvh_build_host(s)
    assert len(nodes) > s.num_triangles and depth > 10


REF_SCENES = "/root/reference/src/assets/scenes"


@pytest.mark.skipif(not os.path.exists(REF_SCENES), reason="reference assets are only present in the build container")
def test_reference_file_scene_factories():
    """createKnobScene / createShaderBallScene[WithLight] (scene.cpp:211-298) on the reference's own OBJ files:
    mesh count, hard-coded cameras / materials / light, a valid BVH and a finite oracle frame."""
    import oracle_lib as ol
    knob = scenes.knob_scene(os.path.join(REF_SCENES, "knob.obj"))
    assert [m["name"] for m in knob.meshes][:4] == ["background", "inner", "logo", "outer"]
    assert knob.num_triangles == 11970 and knob.camera["yfov"] == 15.0
    kinds = [knob.materials[m["material"]]["type"] for m in knob.meshes[:4]]
    assert kinds == [scenes.DIFFUSE, scenes.CONDUCTOR, scenes.CONDUCTOR, scenes.DIELECTRIC]
    ball = scenes.shaderball_scene(os.path.join(REF_SCENES, "shaderball", "shaderball.obj"), with_light=True)
    assert len(ball.meshes) >= 4 and ball.materials[ball.meshes[3]["material"]]["type"] == scenes.CONDUCTOR
    assert len(ball.lights) == 1 and ball.lights[0]["type"] == scenes.DISTANT and tuple(ball.lights[0]["position"]) == (0.0, -1.0, 0.0)
    for s in (knob, ball):
        nodes, _, depth = jtx.api.bvh_build_host(s)
        assert len(nodes) > s.num_triangles
        acc, img, cnt = ol.OracleScene(s).render(s.camera_desc(48, 32, 1, 1, 4), threads=4)
        assert np.isfinite(acc).all() and cnt["n_closest"] >= 48 * 32 and img.any()
    helmet = scenes.create_scene(os.path.join(REF_SCENES, "helmet.glb"), background=(0, 0, 0))
    assert helmet.sky == (0, 0, 0) and helmet.num_triangles == 15452


def test_create_scene_transform_applies_to_first_mesh_only(tmp_path):
    src, _ = _source_scene()
    for m in src.materials:
        m["albedo_tex"] = m["mr_tex"] = -1
    path = str(tmp_path / "two.glb")
    gltf.write_glb(path, src)
    t = np.eye(4, dtype=np.float32); t[:3, 3] = (1.0, 2.0, 3.0); t[0, 0] = 2.0
    s = scenes.create_scene(path, transform=t, background=(0.1, 0.1, 0.1))
    assert np.array_equal(s.meshes[0]["vertices"], (src.meshes[0]["vertices"] * np.float32([2, 1, 1]) + np.float32([1, 2, 3])).astype(np.float32))
    assert np.array_equal(s.meshes[1]["vertices"], src.meshes[1]["vertices"])          # scene.cpp:203-206 touches meshes[0] only


def test_external_uris_must_stay_inside_the_asset_directory(tmp_path):
    """A .gltf is untrusted input: absolute paths, '..' escapes and URL schemes in buffer / image URIs are refused."""
    import json
    (tmp_path / "a").mkdir()
    (tmp_path / "secret.bin").write_bytes(b"\0" * 16)
    (tmp_path / "a" / "ok.bin").write_bytes(b"\0" * 16)
    for uri, ok in (("ok.bin", True), ("./ok.bin", True), ("../secret.bin", False), (str(tmp_path / "secret.bin"), False),
                    ("file:///etc/passwd", False), ("%2e%2e/secret.bin", False)):
        doc = {"asset": {"version": "2.0"}, "buffers": [{"uri": uri, "byteLength": 16}]}
        p = tmp_path / "a" / "s.gltf"
        p.write_text(json.dumps(doc))
        if ok:
            gltf._external(str(tmp_path / "a"), uri)
        else:
            with pytest.raises(ValueError):
                gltf._external(str(tmp_path / "a"), uri)


def test_camera_save_writes_png_and_ppm(tmp_path):
    """RGB8Image::save (image.cpp:11-25) of the Python mirror: rows flipped; .png decodes back to the image"""
    import jtx_pathtracer_amd as jtx
    cam = jtx.StaticCamera.__new__(jtx.StaticCamera)
    cam.width_, cam.height_ = 23, 11
    cam.img_ = (np.arange(23 * 11 * 3) % 251).astype(np.uint8).reshape(11, 23, 3)
    cam.save(str(tmp_path / "o.png")); cam.save(str(tmp_path / "o.ppm"))
    assert np.array_equal(gltf.decode_png((tmp_path / "o.png").read_bytes()), cam.img_[::-1])
    ppm = (tmp_path / "o.ppm").read_bytes()
    assert ppm.startswith(b"P6\n23 11\n255\n") and ppm[len(b"P6\n23 11\n255\n"):] == cam.img_[::-1].tobytes()
