"""In-tree build of libjtx_mi.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libjtx_mi.so")
SOURCES = ["jtx_kernels.hip", "jtx_alt.hip", "jtx_wavefront.hip", "jtx_capi.hip", "jtx_multi.hip", "jtx_refit.hip", "jtx_build_dev.hip", "jtx_bvh_build.cpp", "jtx_jpeg.cpp", "jtx_exr.cpp", "jtx_png.cpp"]
HEADERS = ["jtx_device_math.hpp", "jtx_bxdf.hpp", "jtx_scene_dev.hpp", "jtx_launch.hpp", "jtx_host.hpp", "jtx_tiles.hpp", "jtx_wide_quant.hpp", "jtx_inflate.hpp", "jtx_profile.hpp", "jtx_profile_readers.hpp", "jtx_progressive.hpp",
           os.path.join("..", "..", "include", "jtx_mi.h")]
# -ffp-contract=off: device results must equal the strict-fp32 CPU oracle bit for bit.
# -fno-slp-vectorize: the SLP vectoriser turns pairs of fp32 operations into v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32.  On gfx950 a packed
# fp32 instruction takes the two issue slots of the operations it replaces, lets nothing run beside it on the half-rate pipe, and wants its
# operands in aligned register pairs (profiles/r05_box_rates.txt): without it the path kernels spill 12 / 50 / 53 VGPRs instead of 45 / 86 /
# 160 and run 11 % (C2), 5 % (C3), 8 % (C5) faster.  The arithmetic is the same operation for operation (packed or not, each lane is IEEE fp32).
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-std=c++17",
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


def _compile_one(args):
    cc, src, obj, flags = args
    cmd = [cc] + flags + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    return src, cmd, r.returncode, r.stdout + r.stderr


TEST_LIB = os.path.join(_HERE, "libjtx_mi_testhooks.so")


def build_test_hooks(force=False):
    """jtx-pathtracer_amd/libjtx_mi_testhooks.so: the product's objects with jtx_capi.hip recompiled -DJTX_TEST_HOOKS (the fault injection
    of test_device_rebuild_is_failure_atomic).  Test infrastructure: nothing in the product loads it; the tests name it through JTX_MI_LIB
    in a child process."""
    import hashlib
    build_all(force=False)
    cc = _hipcc()
    cflags = [f for f in FLAGS if f != "-shared"] + os.environ.get("JTX_EXTRA_HIPCC_FLAGS", "").split()
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    tag = hashlib.sha1(" ".join(cflags).encode()).hexdigest()[:10]
    objs = [os.path.join(objdir, f"{os.path.splitext(src)[0]}.{tag}.o") for src in SOURCES]
    # the product's objects of THIS flag set must stand: a fresh libjtx_mi.so says nothing about them (build/ does not travel to the
    # GPU box, JTX_EXTRA_HIPCC_FLAGS moves the tag) -- compile what is missing or stale
    if not all(os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(os.path.join(CSRC, src)) for o, src in zip(objs, SOURCES)):
        build_all(force=True)
    hooked = {}                                    # sources with test hooks (fault injection of the rebuild; a refused peer access): recompiled -DJTX_TEST_HOOKS
    for hsrc in ("jtx_capi.hip", "jtx_multi.hip"):
        hook = os.path.join(objdir, f"{os.path.splitext(hsrc)[0]}.testhooks.{tag}.o")
        newest = max(os.path.getmtime(os.path.join(CSRC, d)) for d in HEADERS + [hsrc] if os.path.exists(os.path.join(CSRC, d)))
        if force or not os.path.exists(hook) or os.path.getmtime(hook) < newest:
            src, cmd, rc, out = _compile_one((cc, hsrc, hook, cflags + ["-DJTX_TEST_HOOKS=1"]))
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {hsrc} (test hooks):\n" + out)
        hooked[os.path.splitext(hsrc)[0] + "."] = hook
    if force or not os.path.exists(TEST_LIB) or os.path.getmtime(TEST_LIB) < max([os.path.getmtime(h) for h in hooked.values()] + [os.path.getmtime(LIB)]):
        objs = [next((h for k, h in hooked.items() if os.path.basename(o).startswith(k)), o) for o in objs]
        r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", TEST_LIB] + objs, cwd=CSRC, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc (link, test hooks) failed:\n" + r.stdout + r.stderr)
    return TEST_LIB


def build_all(force=False, verbose=False):
    """Compile every HIP source into jtx-pathtracer_amd/libjtx_mi.so.  Returns the library path.
    One object per source under jtx-pathtracer_amd/build/ (compiled in parallel, re-used while neither the source, a header nor
    the flags changed), then one link; JTX_BUILD_JOBS sets the number of compilers at a time."""
    if not force and lib_is_built():
        return LIB
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    cc = _hipcc()
    cflags = [f for f in FLAGS if f != "-shared"] + os.environ.get("JTX_EXTRA_HIPCC_FLAGS", "").split()
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    tag = hashlib.sha1(" ".join(cflags).encode()).hexdigest()[:10]
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, f"{os.path.splitext(src)[0]}.{tag}.o")
        objs.append(obj)
        fresh = os.path.exists(obj) and os.path.getmtime(obj) >= max(hdr_t, os.path.getmtime(os.path.join(CSRC, src)))
        if force or not fresh:
            jobs.append((cc, src, obj, cflags))
    nj = int(os.environ.get("JTX_BUILD_JOBS", "0")) or min(6, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=max(1, nj)) as ex:
        for src, cmd, rc, out in ex.map(_compile_one, jobs):
            if verbose:
                print(" ".join(cmd))
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n" + out)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc (link) failed:\n" + r.stdout + r.stderr)
    return LIB
