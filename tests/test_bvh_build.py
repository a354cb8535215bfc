"""Host-only invariants of the BVH builder (csrc/bvh_build.h): compiled with hipcc for the host and
run as a plain program - no GPU (see tests/cpp/test_bvh_build.cpp for what is checked)."""
import os
import shutil
import subprocess

import pytest

from conftest import REPO


def test_bvh_builder_invariants(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = tmp_path / "test_bvh_build"
    cmd = [hipcc, "-std=c++17", "-O1", "--offload-arch=gfx950", "-I", os.path.join(REPO, "path-tracer_amd", "csrc"),
           "-I", os.path.join(REPO, "include"), os.path.join(REPO, "tests", "cpp", "test_bvh_build.cpp"), "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.strip() == "ok", run.stdout + run.stderr
