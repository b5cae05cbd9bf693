#!/usr/bin/env python3
"""Generates tests/golden/png_cases.npz: small PNG files (written with Pillow) and what the REFERENCE's decoder makes of them --
stbi_load_from_memory of ext/stb/stb_image.h through oracle/_ref (container only).  gltf.decode_png must reproduce the bytes:
colour types 0 / 2 / 3 / 4 / 6, 1 / 2 / 4 / 8 / 16 bits, palettes with and without tRNS, colour-key transparency, Adam7
interlacing, every filter type.  Run in the build container:  python tests/golden/make_png_golden.py"""
import io
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_jpeg_golden import ref_stb, ref_decode, picture  # noqa: E402
from jtx_pathtracer_amd import gltf  # noqa: E402


def png(img, **kw):
    buf = io.BytesIO(); img.save(buf, "PNG", **kw); return buf.getvalue()


def cases():
    out = []
    rgb = picture(37, 23, 3)
    out.append(("rgb8", png(Image.fromarray(rgb))))
    out.append(("rgb8_adam7", png(Image.fromarray(picture(19, 14, 4)))))                      # replaced below by an interlaced copy
    rgba = np.dstack([picture(29, 17, 5), picture(29, 17, 6)[..., :1]])
    out.append(("rgba8", png(Image.fromarray(rgba, "RGBA"))))
    out.append(("grey8", png(Image.fromarray(picture(31, 9, 7)[..., 0], "L"))))
    out.append(("greyalpha8", png(Image.fromarray(np.dstack([picture(15, 21, 8)[..., 0], picture(15, 21, 9)[..., 1]]), "LA"))))
    g16 = (np.random.RandomState(1).rand(13, 27) * 65535).astype(np.uint16)
    out.append(("grey16", png(Image.fromarray(g16, "I;16"))))
    pal = Image.fromarray(rgb).quantize(colors=23)
    out.append(("palette8", png(pal)))
    out.append(("palette4", png(Image.fromarray(rgb).quantize(colors=11), bits=4)))
    out.append(("palette2", png(Image.fromarray(rgb).quantize(colors=4), bits=2)))
    out.append(("palette1", png(Image.fromarray(rgb).quantize(colors=2), bits=1)))
    palt = Image.fromarray(rgb).quantize(colors=16)
    out.append(("palette_trns", png(palt, transparency=bytes([0, 64, 128, 255, 10]))))
    out.append(("grey1", png(Image.fromarray(picture(33, 10, 10)[..., 0] > 128))))
    out.append(("rgb_colorkey", png(Image.fromarray(np.where((np.arange(37 * 23).reshape(23, 37, 1) % 5) == 0, np.uint8(200), rgb).astype(np.uint8)), transparency=(200, 200, 200))))
    out.append(("grey_colorkey", png(Image.fromarray((picture(20, 12, 11)[..., 0] // 32 * 32).astype(np.uint8), "L"), transparency=96)))
    out.append(("rgb8_1x1", png(Image.fromarray(picture(1, 1, 12)))))
    out.append(("rgb8_big_paeth", png(Image.fromarray(picture(96, 64, 13)), compress_level=9)))
    return out


def interlace(data):
    """re-encode a non-interlaced 8-bit PNG as Adam7 (Pillow cannot write interlaced files)"""
    import struct, zlib
    px = gltf.decode_png(data)
    h, w, c = px.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    raw = b""
    for (x0, y0, dx, dy) in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
        sub = px[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        for row in sub:
            raw += b"\0" + row.tobytes()

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")


def main():
    lib = ref_stb()
    store = {}
    for name, data in cases():
        if name.endswith("adam7"):
            data = interlace(data)
        store[name + ".png"] = np.frombuffer(data, np.uint8)
        store[name + ".u8"] = ref_decode(lib, data)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "png_cases.npz"), **store)
    bad = []
    for name, _ in cases():
        try:
            got = gltf.decode_png(store[name + ".png"].tobytes())
            if got.shape != store[name + ".u8"].shape or not np.array_equal(got, store[name + ".u8"]):
                bad.append((name, got.shape, store[name + ".u8"].shape))
        except Exception as e:
            bad.append((name, repr(e)))
    print(f"{len(cases())} files; the package's reader differs on: {bad}")


if __name__ == "__main__":
    main()
