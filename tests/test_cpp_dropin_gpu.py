"""Runs the C++ restatement of the reference's hot-path unit tests against the GPU-backed
drop-in headers (tests/cpp/test_dropin.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_reference_unit_tests_on_the_dropin():
    exe = os.path.join(ROOT, "tests", "cpp", "test_dropin")
    assert os.path.exists(exe), "tests/cpp/test_dropin is not built (make)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
