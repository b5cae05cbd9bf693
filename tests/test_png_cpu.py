"""PNG ingestion (image.cpp:70,97 -> stbi_loadf): gltf.decode_png against the REFERENCE's own decoder.  tests/golden/png_cases.npz
holds small PNG files and their decodes by ext/stb/stb_image.h (tests/golden/make_png_golden.py through oracle/_ref): colour
types 0 / 2 / 3 / 4 / 6, 1 - 16 bits, palettes, tRNS (palette alpha and colour keys), Adam7, every filter.  No GPU."""
import os

import numpy as np
import pytest

from jtx_pathtracer_amd import gltf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "png_cases.npz"))
NAMES = sorted(k[:-4] for k in GOLD.files if k.endswith(".png"))


@pytest.mark.parametrize("name", NAMES)
def test_png_decode_equals_the_reference_decoder(name):
    got = gltf.decode_png(GOLD[name + ".png"].tobytes())
    want = GOLD[name + ".u8"]
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(got, want), f"{name}: {(got != want).sum()} samples differ"


def test_cases_cover_the_format():
    assert len(NAMES) >= 16
    import struct
    seen = set()
    for n in NAMES:
        d = GOLD[n + ".png"].tobytes()
        w, h, depth, ctype, _, _, il = struct.unpack(">IIBBBBB", d[16:29])
        seen.add((ctype, depth, il, b"tRNS" in d))
    assert {c for c, _, _, _ in seen} == {0, 2, 3, 4, 6}
    assert {d for _, d, _, _ in seen} >= {1, 2, 4, 8, 16}
    assert any(il for _, _, il, _ in seen) and any(t and c in (0, 2) for c, _, _, t in seen) and any(t and c == 3 for c, _, _, t in seen)


def test_bad_png_is_an_error():
    ok = GOLD["rgb8.png"].tobytes()
    for bad in (b"", ok[:20], ok[:8] + ok[33:], ok[:-40], ok.replace(b"IDAT", b"IDAX")):
        with pytest.raises(Exception):
            gltf.decode_png(bad)
