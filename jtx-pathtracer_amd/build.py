"""In-tree build of libjtx_mi.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libjtx_mi.so")
SOURCES = ["jtx_kernels.hip", "jtx_alt.hip", "jtx_wavefront.hip", "jtx_capi.hip", "jtx_multi.hip", "jtx_refit.hip", "jtx_build_dev.hip", "jtx_bvh_build.cpp", "jtx_jpeg.cpp", "jtx_exr.cpp", "jtx_png.cpp"]
HEADERS = ["jtx_device_math.hpp", "jtx_bxdf.hpp", "jtx_scene_dev.hpp", "jtx_launch.hpp", "jtx_host.hpp", "jtx_tiles.hpp", "jtx_wide_quant.hpp", "jtx_inflate.hpp",
           os.path.join("..", "..", "include", "jtx_mi.h")]
# -ffp-contract=off: device results must equal the strict-fp32 CPU oracle bit for bit.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Wall", "-Wno-unused-function"]


# (experiment kernels that lost -- DESIGN.md section 10 -- live in tools/experiments/, outside the library)


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def lib_is_built():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return all(os.path.getmtime(d) <= t for d in deps if os.path.exists(d))


def build_all(force=False, verbose=False):
    """Compile every HIP source into jtx-pathtracer_amd/libjtx_mi.so.  Returns the library path."""
    if not force and lib_is_built():
        return LIB
    cmd = [_hipcc()] + FLAGS + os.environ.get("JTX_EXTRA_HIPCC_FLAGS", "").split() + ["-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return LIB
