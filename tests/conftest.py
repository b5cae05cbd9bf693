import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the runner captures file descriptor 2, so the C++ runtime's last words before an abort are lost: the library writes them
    # (reason + native backtrace) here instead (jtx_capi.hip, JTX_ABORT_LOG)
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        os.environ.setdefault("JTX_ABORT_LOG", os.path.join(out, "jtx_abort.log"))
    except OSError:
        pass
    # A fresh checkout running the CPU suite (-m "not gpu"): the host-side entry points these tests call (decoders, BVH builders, the
    # export table) live in libjtx_mi.so and the oracle in oracle/_build -- build what is missing, as __graft_entry__.build() does.
    # Never for the GPU tests: there a missing library is a failure, not something to paper over (fixture `gpu`).
    if "not gpu" in (config.getoption("markexpr") or ""):
        try:
            import jtx_pathtracer_amd as jtx
            if not jtx.lib_is_built() or not os.path.exists(os.path.join(ROOT, "oracle", "_build", "libjtx_oracle.so")):
                import __graft_entry__
                __graft_entry__.build()
        except Exception as e:                                # (the tests that need the library then say what is missing)
            print("conftest: build of a fresh checkout failed:", e, file=sys.stderr)


def _has_gpu():
    try:
        import jtx_pathtracer_amd as jtx
        import ctypes as C
        lib = jtx._capi.load()
        n = C.c_int32(0)
        return lib.jtx_mi_device_count(C.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """The product library on a GPU box.  GPU tests FAIL (not skip) when the HIP library is missing."""
    # torch (device buffers of the *_device tests) brings its own HIP runtime: let it initialise first, as in
    # bench.py, so that any test subset sees the same load order as the whole suite
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except ImportError:
        pass
    import jtx_pathtracer_amd as jtx
    jtx._capi.load()            # raises if libjtx_mi.so is absent: no silent fallback
    if not _has_gpu():
        pytest.fail("no HIP device visible: -m gpu tests need an MI355X")
    return jtx


@pytest.fixture(scope="session")
def cornell_pair(gpu):
    import oracle_lib as ol
    data = gpu.scenes.cornell()
    sc = gpu.Scene(data)
    sc.buildBVH()
    return data, sc, ol.OracleScene(data)


@pytest.fixture(scope="session")
def mixed_pair(gpu):
    import oracle_lib as ol
    data = gpu.scenes.mixed(sphere_res=(24, 12))
    sc = gpu.Scene(data)
    sc.buildBVH()
    return data, sc, ol.OracleScene(data)
