"""The host BVH builder under sanitizers (SURVEY section 5 lists race detection among the reference's missing aids; GPU
sanitizers are not available on the pool, the builder is plain host C++): ThreadSanitizer on the parallel build, AddressSanitizer
+ UBSan on the same program; the 8-thread tree must equal the 1-thread tree byte for byte.  No GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "cpp", "bvh_sanitize.cpp"), os.path.join(ROOT, "jtx-pathtracer_amd", "csrc", "jtx_bvh_build.cpp")]


@pytest.mark.parametrize("flags,ntri", [(["-fsanitize=thread"], 60000), (["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], 60000)])
def test_parallel_bvh_build_is_clean_under_sanitizers(tmp_path, flags, ntri):
    exe = str(tmp_path / "bvh_sanitize")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g"] + flags + ["-o", exe] + SRC + ["-lpthread"], check=True)
    r = subprocess.run([exe, str(ntri)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "bad-input errors 3" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
