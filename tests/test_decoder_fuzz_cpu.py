"""The JPEG and EXR readers under AddressSanitizer + UBSan on the host, fed corrupted copies of the golden files: scene assets
are untrusted input.  (GPU sanitizers are not available on the pool; these two readers are plain host C++.)  No GPU."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_image_readers_survive_corrupted_files_under_sanitizers(tmp_path):
    exe = str(tmp_path / "decoder_fuzz")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "decoder_fuzz.cpp"), os.path.join(ROOT, "jtx-pathtracer_amd", "csrc", "jtx_jpeg.cpp"),
                    os.path.join(ROOT, "jtx-pathtracer_amd", "csrc", "jtx_exr.cpp"),
                    os.path.join(ROOT, "jtx-pathtracer_amd", "csrc", "jtx_png.cpp")], check=True)
    files = []
    for gold, ext in (("jpeg_cases.npz", ".jpg"), ("exr_cases.npz", ".exr"), ("png_cases.npz", ".png")):
        g = np.load(os.path.join(ROOT, "tests", "golden", gold))
        for k in sorted(g.files):
            if k.endswith(ext):
                p = tmp_path / k
                p.write_bytes(g[k].tobytes())
                files.append(str(p))
    assert len(files) >= 50
    r = subprocess.run([exe, "200"] + files, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    decoded, refused = [int(x) for x in r.stdout.replace(",", "").split() if x.isdigit()]
    assert decoded > 200 and refused > 200          # both outcomes occur; neither is a crash
