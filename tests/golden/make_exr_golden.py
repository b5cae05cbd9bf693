#!/usr/bin/env python3
"""Generates tests/golden/exr_cases.npz: small OpenEXR files (written here by gltf.encode_exr from synthetic pixels) and what
the REFERENCE's reader makes of them -- LoadEXRFromMemory of ext/tinyexr/tinyexr.h over stb's zlib, compiled where it lies by
`make -C oracle ref` and configured as src/image.cpp:1-9 configures it (container only) -- plus the CRC-32 of its output for
the eleven maps under assets/scenes/shaderball/maps.  The product's reader (csrc/jtx_exr.cpp) must reproduce every float bit
for bit.  Run in the build container:  python tests/golden/make_exr_golden.py"""
import ctypes as C
import glob
import os
import subprocess
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jtx_pathtracer_amd import gltf  # noqa: E402

MAPS = "/root/reference/src/assets/scenes/shaderball/maps"


def ref_exr():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, capture_output=True)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_exr.so"))
    lib.ref_exr_from_memory.restype = C.c_int
    lib.ref_exr_from_memory.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p)]
    lib.ref_exr_free.argtypes = [C.POINTER(C.c_float)]
    return lib


def ref_encode(lib, img, fp16, compression):
    """the reference's own writer (SaveEXRImageToMemory through oracle/ref_exr.cpp): compression 4 = PIZ"""
    img = np.ascontiguousarray(img, np.float32)
    h, w, c = img.shape
    lib.ref_exr_save.restype = C.c_size_t
    lib.ref_exr_save.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_char_p)]
    out = C.POINTER(C.c_ubyte)(); err = C.c_char_p()
    n = lib.ref_exr_save(img.ctypes.data_as(C.POINTER(C.c_float)), w, h, c, fp16, compression, C.byref(out), C.byref(err))
    if n == 0:
        raise RuntimeError(f"tinyexr could not write the file: {err.value}")
    data = bytes(out[:n])
    lib.ref_exr_free_bytes.argtypes = [C.POINTER(C.c_ubyte)]; lib.ref_exr_free_bytes(out)
    return data


def ref_decode(lib, data):
    p = C.POINTER(C.c_float)(); w, h = C.c_int(), C.c_int(); err = C.c_char_p()
    rc = lib.ref_exr_from_memory(data, len(data), C.byref(p), C.byref(w), C.byref(h), C.byref(err))
    if rc != 0:
        raise RuntimeError(f"tinyexr refused the file: {rc} {err.value}")
    a = np.ctypeslib.as_array(p, shape=(h.value * w.value * 4,)).copy().reshape(h.value, w.value, 4)
    lib.ref_exr_free(p)
    return a


def picture(w, h, c, seed):
    rs = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([0.5 + 0.5 * np.sin(x / 5.0 + seed), 2.0 ** ((y - h / 2) / 3.0), ((x // 7 + y // 5) % 2) * 3.5, 0.25 + 0.5 * rs.rand(h, w)], -1)[..., :c]
    img = img + rs.normal(0, 0.05, img.shape)
    img[0, 0, 0] = 0.0; img[-1, -1, 0] = 6.0e-6          # a half denormal
    if h > 2 and w > 2:
        img[1, 1, 0] = 70000.0                           # overflows HALF to +inf
        img[2, 1, 0] = -1.5
    return img.astype(np.float32)


def cases():
    out = []
    for name, (w, h, c), kw in [
        ("zip_half_rgb_40x37", (40, 37, 3), dict(compression="zip")),
        ("zip_half_rgba_33x16", (33, 16, 4), dict(compression="zip")),
        ("zip_half_rgb_1x1", (1, 1, 3), dict(compression="zip")),
        ("zip_float_rgb_21x19", (21, 19, 3), dict(compression="zip", half=False)),
        ("zips_half_rgb_50x9", (50, 9, 3), dict(compression="zips")),
        ("rle_half_rgb_64x20", (64, 20, 3), dict(compression="rle")),
        ("none_half_rgba_17x5", (17, 5, 4), dict(compression="none")),
        ("none_float_grey_12x7", (12, 7, 1), dict(compression="none", half=False)),
        ("zip_half_grey_30x33", (30, 33, 1), dict(compression="zip")),
        ("zip_half_rgb_decreasing_y_24x40", (24, 40, 3), dict(compression="zip", line_order=1)),
        ("zip_half_rgb_window_19x23", (19, 23, 3), dict(compression="zip", origin=(-7, 12))),
        ("zip_half_noise_stored_16x16", (16, 16, 3), dict(compression="zip")),
    ]:
        img = picture(w, h, c, len(out) + 1)
        if "noise" in name:                              # incompressible: the block is stored as is
            img = np.random.RandomState(5).rand(h, w, c).astype(np.float32) * 100
        if "rle" in name:
            img[:, : w // 2] = 0.75                        # long runs
        out.append((name, gltf.encode_exr(img, **kw)))
    return out


def piz_cases(lib):
    """PIZ files written by the reference's tinyexr: 32-line blocks, Huffman + wavelet + value table"""
    out = []
    rs = np.random.RandomState(11)
    smooth = picture(70, 45, 3, 3)
    out.append(("piz_half_rgb_70x45", ref_encode(lib, smooth, 1, 4)))                     # two blocks, odd sizes, 14-bit wavelet
    out.append(("piz_half_rgba_33x64", ref_encode(lib, picture(33, 64, 4, 5), 1, 4)))
    out.append(("piz_half_grey_19x7", ref_encode(lib, picture(19, 7, 1, 6), 1, 4)))
    out.append(("piz_float_rgb_noise_200x40", ref_encode(lib, (rs.rand(40, 200, 3) * 1000).astype(np.float32), 0, 4)))   # > 2^14 distinct words: modulo wavelet
    out.append(("piz_half_zero_40x33", ref_encode(lib, np.zeros((33, 40, 3), np.float32), 1, 4)))       # empty bitmap
    flat = np.full((36, 48, 3), 0.25, np.float32); flat[10:20, 5:30] = 3.0
    out.append(("piz_half_flat_48x36", ref_encode(lib, flat, 1, 4)))                      # long runs: the run-length symbol
    out.append(("piz_half_rgb_1x1", ref_encode(lib, picture(1, 1, 3, 7), 1, 4)))
    out.append(("piz_float_rgb_3x50", ref_encode(lib, picture(3, 50, 3, 8), 0, 4)))
    return out


def main():
    lib = ref_exr()
    store = {}
    for name, data in cases() + piz_cases(lib):
        store[name + ".exr"] = np.frombuffer(data, np.uint8)
        store[name + ".f32"] = ref_decode(lib, data)
        mine = gltf.decode_exr(data)
        assert np.array_equal(mine.view(np.uint32), store[name + ".f32"].view(np.uint32)), name
    maps = {}
    for f in sorted(glob.glob(os.path.join(MAPS, "*.exr"))):
        a = ref_decode(lib, open(f, "rb").read())
        maps[os.path.basename(f)] = [a.shape[0], a.shape[1], zlib.crc32(a.tobytes())]
    store["reference_maps.json"] = np.frombuffer(repr(maps).encode(), np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "exr_cases.npz"), **store)
    print(f"{sum(k.endswith('.exr') for k in store)} files, {len(maps)} reference maps, {sum(v.nbytes for v in store.values())} bytes")


if __name__ == "__main__":
    main()
