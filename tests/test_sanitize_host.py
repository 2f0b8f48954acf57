"""Host-only code under AddressSanitizer + UBSan (CPU; the GPU pool runs no sanitizers): PLY ingest incl. 1,500 byte- and
header-level mutations of the fixture files, mesh refinement / transform / append, the BVH builder on real and degenerate
inputs, every scene preset, the PPM / PFM writers (tests/sanitize_host.cpp).  Any report aborts the binary."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_code_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_host")
    csrc = os.path.join(ROOT, "parallelraytracing_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I", os.path.join(ROOT, "include"), "-I", csrc, os.path.join(ROOT, "tests", "sanitize_host.cpp"),
           os.path.join(csrc, "prt_host.cpp"), os.path.join(csrc, "bvh.cpp"), "-pthread", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe, os.path.join(ROOT, "assets", "models"), "1500"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "no sanitizer report" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr
