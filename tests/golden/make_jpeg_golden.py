#!/usr/bin/env python3
"""Generates tests/golden/jpeg_cases.npz: small JPEG files (encoded here with Pillow from synthetic pixels) and what the
REFERENCE's decoder makes of them -- stbi_load_from_memory / stbi_loadf_from_memory of ext/stb/stb_image.h, compiled
where it lies by `make -C oracle ref` (container only).  The product's decoder (csrc/jtx_jpeg.cpp) must reproduce the
bytes; gltf.ldr_to_float must reproduce the floats.  Run in the build container:  python tests/golden/make_jpeg_golden.py"""
import ctypes as C
import io
import os
import subprocess
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ref_stb():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, capture_output=True)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_stb.so"))
    lib.ref_stbi_load_from_memory.restype = C.POINTER(C.c_uint8)
    lib.ref_stbi_load_from_memory.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.ref_stbi_loadf_from_memory.restype = C.POINTER(C.c_float)
    lib.ref_stbi_loadf_from_memory.argtypes = lib.ref_stbi_load_from_memory.argtypes
    lib.ref_stbi_free.argtypes = [C.c_void_p]
    return lib


def ref_decode(lib, data, as_float=False):
    x, y, c = C.c_int(), C.c_int(), C.c_int()
    f = lib.ref_stbi_loadf_from_memory if as_float else lib.ref_stbi_load_from_memory
    p = f(data, len(data), C.byref(x), C.byref(y), C.byref(c))
    if not p:
        raise RuntimeError("stb_image refused the file")
    n = x.value * y.value * c.value
    a = np.ctypeslib.as_array(p, shape=(n,)).copy().reshape(y.value, x.value, c.value)
    lib.ref_stbi_free(p)
    return a


def picture(w, h, seed):
    rs = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([127 + 120 * np.sin(x / 5.0 + seed), 127 + 120 * np.cos(y / 3.0), 255 * ((x // 7 + y // 5) % 2)], -1)
    img += rs.normal(0, 18, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def cases():
    out = []
    for name, (w, h), kw in [
        ("base_420_48x40", (48, 40), dict(subsampling=2)),
        ("base_420_odd_33x17", (33, 17), dict(subsampling=2)),
        ("base_422_50x23", (50, 23), dict(subsampling=1)),
        ("base_444_31x29", (31, 29), dict(subsampling=0)),
        ("base_420_q30_64x64", (64, 64), dict(subsampling=2, quality=30)),
        ("base_420_q97_40x24", (40, 24), dict(subsampling=2, quality=97)),
        ("base_420_restart_57x35", (57, 35), dict(subsampling=2, restart_marker_blocks=3)),
        ("prog_444_45x38", (45, 38), dict(subsampling=0, progressive=True)),
        ("prog_420_63x41", (63, 41), dict(subsampling=2, progressive=True)),
        ("prog_420_q40_80x50", (80, 50), dict(subsampling=2, progressive=True, quality=40)),
        ("base_1x1", (1, 1), dict(subsampling=2)),
        ("base_420_9x1", (9, 1), dict(subsampling=2)),
    ]:
        buf = io.BytesIO()
        Image.fromarray(picture(w, h, len(out) + 1)).save(buf, "JPEG", **{"quality": 85, **kw})
        out.append((name, buf.getvalue()))
    for name, (w, h), kw in [("grey_base_37x21", (37, 21), {}), ("grey_prog_64x40", (64, 40), dict(progressive=True))]:
        buf = io.BytesIO()
        Image.fromarray(picture(w, h, 99)[..., 0]).save(buf, "JPEG", **{"quality": 80, **kw})
        out.append((name, buf.getvalue()))
    return out


def main():
    lib = ref_stb()
    store = {}
    for name, data in cases():
        store[name + ".jpg"] = np.frombuffer(data, np.uint8)
        store[name + ".u8"] = ref_decode(lib, data)
    # stbi_loadf's 8-bit -> float conversion (ldr_to_hdr: pow(v / 255, 2.2)), all 256 levels through the reference's code
    name, data = cases()[0]
    store["loadf.u8"] = ref_decode(lib, data)
    store["loadf.f32"] = ref_decode(lib, data, as_float=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz"), **store)
    print("wrote", len(store), "arrays,", sum(v.nbytes for v in store.values()), "bytes raw")


if __name__ == "__main__":
    main()
