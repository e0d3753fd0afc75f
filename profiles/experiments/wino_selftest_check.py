"""tools/wino_bench.hip (the Toom-Cook F(3,3) / F(4,4) form of the split-f16 conv, an experiment
outside the library; measured on the GPU in round 3 and NOT adopted: profiles/archive/r03/r03_wino_decision.md): it must keep compiling for gfx950, and its GPU-free `selftest quick` --
Toom matrices against direct correlation, a host emulation of the data flow and thread-level host twins of both kernels
through the fp64 checker -- must keep passing."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_wino_bench_builds_and_selftests(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not installed")
    exe = os.path.join(str(tmp_path), "wino_bench")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function",
                           os.path.join(ROOT, "profiles", "experiments", "wino_bench.hip"), "-o", exe])
    r = subprocess.run([exe, "selftest", "quick"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all cases OK" in r.stdout and "MISMATCH" not in r.stdout
    assert r.stdout.count("thread-level twin") >= 3
