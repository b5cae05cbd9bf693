"""glTF 2.0 / GLB ingestion with the semantics of the reference's Assimp import.

`loadScene` (src/loader.cpp:13-225) reads a file through Assimp with
`aiProcess_Triangulate | aiProcess_FlipUVs | aiProcess_GenNormals | aiProcess_PreTransformVertices`
(loader.cpp:21) and `createScene` (src/scene.cpp:176-209) wraps it with a default camera and a
background colour.  Assimp is not available here, so this module restates what that pipeline does to a
glTF file, as far as the hot path can see it:

  * one Mesh per glTF primitive, in scene-graph traversal order, node transforms baked into positions
    (point transform) and normals (inverse-transpose, re-normalised) -- PreTransformVertices;
  * v -> 1 - v on TEXCOORD_0 -- FlipUVs;
  * flat face normals when a primitive has no NORMAL -- GenNormals (vertices are not shared then);
  * one Triangle ref {index, meshIndex} per face, mesh by mesh (loader.cpp:216-222);
  * materials (loader.cpp:104-150): every glTF material carries metallic / roughness factors, so it is
    METALLIC_ROUGHNESS with albedo WHITE, alphaX = metallicFactor, alphaY = roughnessFactor (the
    reference stores the two factors in the alpha fields, loader.cpp:136-137), albedo texture =
    baseColorTexture, metallic-roughness texture = metallicRoughnessTexture; baseColorFactor, emission,
    normal and occlusion maps are ignored exactly as the reference ignores them; a primitive without a
    material gets DIFFUSE (1, 0.3, 0.5) (loader.cpp:206-209);
  * textures (src/image.cpp:59-101 -> stbi_loadf): 8-bit samples become float pow(x / 255, 2.2) per
    colour channel, alpha stays x / 255 (ext/stb/stb_image.h ldr_to_hdr); the file's own channel count
    is kept.  PNG (1-16 bit, grey / RGB / palette, alpha or colour key, Adam7) is decoded here with zlib; JPEG (baseline and
    progressive) and OpenEXR by the library's host decoders (csrc/jtx_jpeg.cpp, csrc/jtx_exr.cpp).

The three decoders are pinned against the reference's own: ext/stb/stb_image.h and ext/tinyexr/tinyexr.h compiled where
they lie by the test infrastructure (tests/golden/{png,jpeg,exr}_cases.npz and their generators).  What remains unpinned is the
Assimp side (mesh splitting, traversal order): covered by round-trip tests against this module's own writer and by
structural checks on the reference's helmet.glb where that file is present.
"""
import json
import os
import struct
import zlib

import numpy as np

from . import scenes

_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


# ------------------------------------------------------------------------------------------------
# PNG (decode for textures, encode for the test writer)
# ------------------------------------------------------------------------------------------------
def decode_png(data):
    """PNG bytes -> uint8 array (H, W, C) as stbi_load_from_memory(.., req_comp = 0) of the reference's stb_image returns it
    (jtx_mi_decode_png, csrc/jtx_png.cpp; pinned by tests/golden/png_cases.npz): C = 1 (grey), 2 (grey + alpha), 3 (RGB /
    palette), 4 (RGBA / palette + tRNS); a tRNS colour key on a grey / RGB image adds the alpha channel; 1 / 2 / 4-bit grey is
    scaled to 0..255; 16-bit samples keep their high byte; Adam7 files are de-interlaced."""
    import ctypes as C
    from . import _capi as capi
    lib = capi.load()
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b"\0")
    w, h, c = C.c_int32(), C.c_int32(), C.c_int32()
    capi.check(lib.jtx_mi_decode_png(buf, len(data), C.byref(w), C.byref(h), C.byref(c), None, 0))
    out = np.zeros((h.value, w.value, c.value), np.uint8)
    capi.check(lib.jtx_mi_decode_png(buf, len(data), C.byref(w), C.byref(h), C.byref(c), out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size))
    return out


def decode_jpeg(data):
    """JPEG bytes -> uint8 (H, W, C), C = 3 or 1: what stbi_load_from_memory(.., req_comp = 0) returns (jtx_mi_decode_jpeg)."""
    import ctypes as C
    from . import _capi as capi
    lib = capi.load()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    w, h, c = C.c_int32(), C.c_int32(), C.c_int32()
    capi.check(lib.jtx_mi_decode_jpeg(buf, len(data), C.byref(w), C.byref(h), C.byref(c), None, 0))
    out = np.zeros((h.value, w.value, c.value), np.uint8)
    capi.check(lib.jtx_mi_decode_jpeg(buf, len(data), C.byref(w), C.byref(h), C.byref(c), out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size))
    return out


def decode_exr(data):
    """OpenEXR bytes -> float32 (H, W, 4), rows top to bottom: what TextureImage::load gets from tinyexr (jtx_mi_decode_exr)."""
    import ctypes as C
    from . import _capi as capi
    lib = capi.load()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    w, h = C.c_int32(), C.c_int32()
    capi.check(lib.jtx_mi_decode_exr(buf, len(data), C.byref(w), C.byref(h), None, 0))
    out = np.zeros((h.value, w.value, 4), np.float32)
    capi.check(lib.jtx_mi_decode_exr(buf, len(data), C.byref(w), C.byref(h), out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
    return out


def load_texture_file(path):
    """TextureImage::load(path) (image.cpp:59-74): .exr through tinyexr -> (H, W, 4) floats as stored; anything else through
    stbi_loadf -> (H, W, C) floats, 8-bit samples raised to 2.2 (PNG and JPEG here).  None when the file cannot be read."""
    try:
        data = open(path, "rb").read()
        ext = path.rsplit(".", 1)[-1]
        if ext in ("exr", "EXR"):
            return decode_exr(data)
        if data[:8] == b"\x89PNG\r\n\x1a\n":
            return ldr_to_float(decode_png(data))
        if data[:2] == b"\xff\xd8":
            return ldr_to_float(decode_jpeg(data))
    except Exception:
        return None
    return None


def encode_exr(img, compression="zip", half=True, names=None, line_order=0, origin=(0, 0)):
    """float32 (H, W, C) -> scan-line OpenEXR bytes (test writer: NONE / RLE / ZIPS / ZIP, HALF or FLOAT channels stored
    in alphabetical order as the format asks; names default to B G R [A] / Y)."""
    import struct, zlib
    img = np.ascontiguousarray(img, np.float32)
    h, w, c = img.shape
    names = names or {1: ["Y"], 3: ["R", "G", "B"], 4: ["R", "G", "B", "A"]}[c]
    order = sorted(range(c), key=lambda i: names[i])
    comp = {"none": 0, "rle": 1, "zips": 2, "zip": 3}[compression]
    lines = 16 if comp == 3 else 1
    x0, y0 = origin

    def attr(name, typ, payload):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(payload)) + payload
    chl = b"".join(names[i].encode() + b"\0" + struct.pack("<iB3xii", 1 if half else 2, 0, 1, 1) for i in order) + b"\0"
    box = struct.pack("<4i", x0, y0, x0 + w - 1, y0 + h - 1)
    hdr = (b"\x76\x2f\x31\x01" + struct.pack("<I", 2) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([comp])) +
           attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", bytes([line_order])) +
           attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) +
           attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    blocks = []
    for r0 in range(0, h, lines):
        rows = img[r0:r0 + lines]
        raw = b"".join((rows[l, :, i].astype(np.float16) if half else rows[l, :, i]).tobytes() for l in range(rows.shape[0]) for i in order)
        if comp == 0:
            data = raw
        else:
            a = np.frombuffer(raw, np.uint8)
            t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
            t[1:] = (t[1:] - t[:-1] + 128 + 256) & 255
            t = t.astype(np.uint8).tobytes()
            if comp == 1:
                out = bytearray(); i = 0
                while i < len(t):                       # ImfRle.cpp: runs of 3+ equal bytes, else literals
                    j = i
                    while j + 1 < len(t) and t[j + 1] == t[i] and j - i < 126:
                        j += 1
                    if j - i >= 2:
                        out += bytes([j - i, t[i]]); i = j + 1
                    else:
                        k = i
                        while k < len(t) and k - i < 127 and not (k + 2 < len(t) and t[k] == t[k + 1] == t[k + 2]):
                            k += 1
                        out += bytes([(256 - (k - i)) & 255]) + t[i:k]; i = k
                data = bytes(out)
            else:
                data = zlib.compress(t, 6)
            if len(data) >= len(raw):
                data = raw
        blocks.append((y0 + r0, data))
    if line_order == 1:
        blocks = blocks[::-1]
    table_at = len(hdr)
    pos = table_at + 8 * len(blocks)
    offs = {}
    body = b""
    for y, data in blocks:
        offs[y] = pos
        body += struct.pack("<iI", y, len(data)) + data
        pos += 8 + len(data)
    table = b"".join(struct.pack("<Q", offs[y0 + r0]) for r0 in range(0, h, lines))
    return hdr + table + body


def encode_png(img):
    """uint8 (H, W, C) -> PNG bytes (filter 0, C in 1..4)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w, c = img.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xffffffff)

    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + \
        chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


def ldr_to_float(px):
    """stbi_loadf on an 8-bit image: pow(x / 255, 2.2) per colour channel, alpha (2nd of 2, 4th of 4) linear."""
    px = np.asarray(px, np.uint8)
    c = px.shape[-1]
    out = np.empty(px.shape, np.float32)
    ncol = c if c in (1, 3) else c - 1
    # (float) (pow(data / 255.0f, stbi__l2h_gamma) * stbi__l2h_scale) with float gamma 2.2f, scale 1.0f (stb_image.h:1573,1869):
    # the quotient is a float, pow runs in double on the promoted operands, the product is rounded to float once
    lut = (np.power((np.arange(256, dtype=np.float32) / np.float32(255.0)).astype(np.float64), np.float64(np.float32(2.2)))
           * np.float64(np.float32(1.0))).astype(np.float32)
    out[..., :ncol] = lut[px[..., :ncol]]
    if ncol < c:
        out[..., ncol:] = px[..., ncol:].astype(np.float32) / np.float32(255.0)
    return out


# ------------------------------------------------------------------------------------------------
# reader
# ------------------------------------------------------------------------------------------------
def _read_container(path):
    with open(path, "rb") as f:
        blob = f.read()
    base = os.path.dirname(os.path.abspath(path))
    if blob[:4] == b"glTF":
        _, _, total = struct.unpack("<4sII", blob[:12])
        pos, doc, bin_chunk = 12, None, None
        while pos < total:
            n, kind = struct.unpack("<II", blob[pos:pos + 8])
            body = blob[pos + 8:pos + 8 + n]
            pos += 8 + n
            if kind == 0x4E4F534A:
                doc = json.loads(body.decode("utf-8"))
            elif kind == 0x004E4942 and bin_chunk is None:
                bin_chunk = body
        return doc, bin_chunk, base
    return json.loads(blob.decode("utf-8")), None, base


def _buffers(doc, bin_chunk, base):
    import base64
    out = []
    for b in doc.get("buffers", []):
        uri = b.get("uri")
        if uri is None:
            out.append(bin_chunk)
        elif uri.startswith("data:"):
            out.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            with open(_external(base, uri), "rb") as f:
                out.append(f.read())
    return out


def _external(base, uri):
    """Path of an external buffer / image URI of a .gltf file: percent-decoded, relative, and inside the asset's own
    directory -- a scene file is untrusted input and must not name /etc/passwd or ../../x."""
    from urllib.parse import unquote
    rel = unquote(uri)
    root = os.path.realpath(base or ".")
    full = os.path.realpath(os.path.join(root, rel))
    if os.path.isabs(rel) or "://" in uri or os.path.commonpath([root, full]) != root:
        raise ValueError(f"glTF external URI escapes the asset directory: {uri!r}")
    return full


def _accessor(doc, bufs, i):
    a = doc["accessors"][i]
    dt, nc = np.dtype(_COMP[a["componentType"]]), _NCOMP[a["type"]]
    count = a["count"]
    if "bufferView" not in a:
        arr = np.zeros((count, nc), dt)
    else:
        v = doc["bufferViews"][a["bufferView"]]
        start = v.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = v.get("byteStride", 0) or dt.itemsize * nc
        buf = bufs[v["buffer"]]
        if stride == dt.itemsize * nc:
            arr = np.frombuffer(buf, dt, count * nc, start).reshape(count, nc)
        else:
            arr = np.stack([np.frombuffer(buf, dt, nc, start + k * stride) for k in range(count)])
    if a.get("normalized") and dt.kind in "ui":
        info = np.iinfo(dt)
        arr = np.maximum(arr.astype(np.float32) / np.float32(info.max), -1.0)
    return arr


def _node_matrix(n):
    if "matrix" in n:
        return np.array(n["matrix"], np.float64).reshape(4, 4).T          # glTF stores column-major
    t = np.array(n.get("translation", (0, 0, 0)), np.float64)
    x, y, z, w = n.get("rotation", (0, 0, 0, 1))
    s = np.array(n.get("scale", (1, 1, 1)), np.float64)
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float64)
    m = np.eye(4)
    m[:3, :3] = r * s[None, :]
    m[:3, 3] = t
    return m


def _image_bytes(doc, bufs, base, img):
    if "bufferView" in img:
        v = doc["bufferViews"][img["bufferView"]]
        s = v.get("byteOffset", 0)
        return bytes(bufs[v["buffer"]][s:s + v["byteLength"]]), img.get("mimeType", "")
    uri = img["uri"]
    if uri.startswith("data:"):
        import base64
        return base64.b64decode(uri.split(",", 1)[1]), uri[5:uri.index(";")]
    with open(_external(base, uri), "rb") as f:
        return f.read(), img.get("mimeType", "")


def load_gltf(path, background=scenes.SKY_BLUE, allow_missing_textures=False):
    """createScene(path, identity, background) (src/scene.cpp:176-209) for a .glb / .gltf file -> SceneData."""
    doc, bin_chunk, base = _read_container(path)
    bufs = _buffers(doc, bin_chunk, base)
    s = scenes.SceneData("File scene")
    s.sky = tuple(background)
    s.camera = dict(center=(0, 0, 8), target=(0, 0, 0), up=(0, 1, 0), yfov=20.0, defocus_angle=0.0, focus_distance=1.0)

    tex_of_image = {}

    def texture_id(texinfo):
        if texinfo is None:
            return -1
        src = doc["textures"][texinfo["index"]].get("source")
        if src is None:
            return -1
        if src in tex_of_image:
            return tex_of_image[src]
        data, mime = _image_bytes(doc, bufs, base, doc["images"][src])
        if data[:8] == b"\x89PNG\r\n\x1a\n":
            px = decode_png(data)
            if px.shape[-1] < 3:
                raise ValueError(f"image {src}: {px.shape[-1]}-channel texture; getTexel (image.hpp:140-153) reads 3 channels")
            s.textures.append(ldr_to_float(px))
            tex_of_image[src] = len(s.textures) - 1
        elif data[:2] == b"\xff\xd8":
            px = decode_jpeg(data)
            if px.shape[-1] < 3:
                raise ValueError(f"image {src}: {px.shape[-1]}-channel texture; getTexel (image.hpp:140-153) reads 3 channels")
            s.textures.append(ldr_to_float(px))
            tex_of_image[src] = len(s.textures) - 1
        elif allow_missing_textures:
            tex_of_image[src] = -1
        else:
            raise ValueError(f"image {src} ({mime or 'unknown type'}): only PNG and JPEG can be decoded here "
                             "(pass allow_missing_textures=True to load the scene without it)")
        return tex_of_image[src]

    mat_index = {}
    for i, m in enumerate(doc.get("materials", [])):                    # loader.cpp:104-150
        name = m.get("name", "")
        if name in mat_index:
            continue                                                     # materialMap.contains(matName)
        pbr = m.get("pbrMetallicRoughness", {})
        s.materials.append(scenes.material(scenes.METALLIC_ROUGHNESS, (1.0, 1.0, 1.0),
                                           alpha_x=float(pbr.get("metallicFactor", 1.0)),
                                           alpha_y=float(pbr.get("roughnessFactor", 1.0)),
                                           albedo_tex=texture_id(pbr.get("baseColorTexture")),
                                           mr_tex=texture_id(pbr.get("metallicRoughnessTexture"))))
        mat_index[name] = len(s.materials) - 1

    def visit(ni, parent):
        n = doc["nodes"][ni]
        m = parent @ _node_matrix(n)
        if "mesh" in n:
            for prim in doc["meshes"][n["mesh"]]["primitives"]:
                if prim.get("mode", 4) != 4:
                    continue                                             # points / lines: dropped by Triangulate + face check
                att = prim["attributes"]
                pos = _accessor(doc, bufs, att["POSITION"]).astype(np.float64)
                idx = _accessor(doc, bufs, prim["indices"]).reshape(-1).astype(np.int64) if "indices" in prim \
                    else np.arange(len(pos), dtype=np.int64)
                idx = idx[: len(idx) // 3 * 3].reshape(-1, 3)
                uv = _accessor(doc, bufs, att["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in att else None
                nrm = _accessor(doc, bufs, att["NORMAL"]).astype(np.float64) if "NORMAL" in att else None
                if nrm is None:                                          # GenNormals: flat, vertices un-shared
                    pos = pos[idx.reshape(-1)]
                    if uv is not None:
                        uv = uv[idx.reshape(-1)]
                    idx = np.arange(len(pos), dtype=np.int64).reshape(-1, 3)
                    fn = np.cross(pos[1::3] - pos[0::3], pos[2::3] - pos[0::3])
                    ln = np.linalg.norm(fn, axis=1, keepdims=True)
                    fn = np.where(ln > 0, fn / np.where(ln > 0, ln, 1), 0)
                    nrm = np.repeat(fn, 3, axis=0)
                wp = (pos @ m[:3, :3].T + m[:3, 3]).astype(np.float32)   # PreTransformVertices
                nm = np.linalg.inv(m[:3, :3]).T
                wn = nrm @ nm.T
                ln = np.linalg.norm(wn, axis=1, keepdims=True)
                wn = np.where(ln > 0, wn / np.where(ln > 0, ln, 1), wn).astype(np.float32)
                if uv is not None:
                    uv = np.stack([uv[:, 0], np.float32(1.0) - uv[:, 1]], 1).astype(np.float32)   # FlipUVs
                if "material" in prim:
                    mat = mat_index[doc["materials"][prim["material"]].get("name", "")]
                else:
                    s.materials.append(scenes.material(scenes.DIFFUSE, (1.0, 0.3, 0.5)))         # loader.cpp:206-209
                    mat = len(s.materials) - 1
                s.add_mesh(idx.astype(np.int32), wp, wn, mat, uvs=uv, name=doc["meshes"][n["mesh"]].get("name", f"mesh_{len(s.meshes)}"))
        for c in n.get("children", []):
            visit(c, m)

    scene_def = doc["scenes"][doc.get("scene", 0)]
    for root in scene_def["nodes"]:
        visit(root, np.eye(4))
    return s


# ------------------------------------------------------------------------------------------------
# writer (tests; also a way to hand a procedural scene to other tools)
# ------------------------------------------------------------------------------------------------
def write_glb(path, data, textures_u8=None, node_matrices=None):
    """SceneData -> .glb.  Geometry is written as given (node matrix = `node_matrices[i]` or identity; mesh
    transforms must be identity); METALLIC_ROUGHNESS / other materials become pbrMetallicRoughness with
    metallicFactor = alpha_x, roughnessFactor = alpha_y; `textures_u8[i]` (H, W, C uint8) is embedded as PNG for
    scene texture i.  v is stored un-flipped (1 - v), so that load_gltf(write_glb(s)) gives s.uvs back."""
    bin_parts, views, accessors = [], [], []

    def add_view(raw, target=None):
        while sum(len(p) for p in bin_parts) % 4:
            bin_parts.append(b"\x00")
        off = sum(len(p) for p in bin_parts)
        bin_parts.append(raw)
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(raw)}
        if target:
            v["target"] = target
        views.append(v)
        return len(views) - 1

    def add_acc(arr, ctype, kind, target, minmax=False):
        arr = np.ascontiguousarray(arr)
        a = {"bufferView": add_view(arr.tobytes(), target), "componentType": ctype, "count": int(arr.shape[0]), "type": kind}
        if minmax:
            a["min"] = [float(x) for x in arr.min(0)]
            a["max"] = [float(x) for x in arr.max(0)]
        accessors.append(a)
        return len(accessors) - 1

    images, textures = [], []
    for t in (textures_u8 or []):
        images.append({"bufferView": add_view(encode_png(t)), "mimeType": "image/png"})
        textures.append({"source": len(images) - 1})
    materials = []
    for i, m in enumerate(data.materials):
        pbr = {"metallicFactor": float(m["alpha_x"]), "roughnessFactor": float(m["alpha_y"])}
        if m["albedo_tex"] >= 0:
            pbr["baseColorTexture"] = {"index": int(m["albedo_tex"])}
        if m["mr_tex"] >= 0:
            pbr["metallicRoughnessTexture"] = {"index": int(m["mr_tex"])}
        materials.append({"name": f"material_{i}", "pbrMetallicRoughness": pbr})
    meshes, nodes = [], []
    for i, m in enumerate(data.meshes):
        att = {"POSITION": add_acc(m["vertices"].astype(np.float32), 5126, "VEC3", 34962, True),
               "NORMAL": add_acc(m["normals"].astype(np.float32), 5126, "VEC3", 34962)}
        if m["uvs"] is not None:
            uv = np.stack([m["uvs"][:, 0], np.float32(1.0) - m["uvs"][:, 1]], 1).astype(np.float32)
            att["TEXCOORD_0"] = add_acc(uv, 5126, "VEC2", 34962)
        prim = {"attributes": att, "indices": add_acc(m["indices"].reshape(-1).astype(np.uint32), 5125, "SCALAR", 34963),
                "material": int(m["material"]), "mode": 4}
        meshes.append({"name": m["name"] or f"mesh_{i}", "primitives": [prim]})
        node = {"mesh": i}
        if node_matrices is not None and node_matrices[i] is not None:
            node["matrix"] = [float(x) for x in np.asarray(node_matrices[i], np.float64).T.reshape(-1)]
        nodes.append(node)
    bin_blob = b"".join(bin_parts)
    bin_blob += b"\x00" * (-len(bin_blob) % 4)
    doc = {"asset": {"version": "2.0", "generator": "jtx-mi"}, "scene": 0, "scenes": [{"nodes": list(range(len(nodes)))}],
           "nodes": nodes, "meshes": meshes, "materials": materials, "accessors": accessors, "bufferViews": views,
           "buffers": [{"byteLength": len(bin_blob)}]}
    if images:
        doc["images"], doc["textures"] = images, textures
    js = json.dumps(doc, separators=(",", ":")).encode("utf-8")
    js += b" " * (-len(js) % 4)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(bin_blob)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(bin_blob), 0x004E4942) + bin_blob)
