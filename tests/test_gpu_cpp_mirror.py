"""GPU: the C++ host mirror (bio_amd/csrc/sketches.hpp) runs the reference's own Go test cases."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_mirror_of_reference_tests():
    exe = os.path.join(ROOT, "bio_amd", "csrc", "test_sketches")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "test_sketches"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "all C++ mirror checks passed" in out.stdout, out.stdout + out.stderr
