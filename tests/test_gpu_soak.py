"""A short, fixed-seed run of the randomised soaks (tests/soak/) as part of the GPU suite."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("script,seconds", [("gpu_fuzz.py", 12), ("gpu_fuzz_symbols.py", 8)])
def test_short_soak(script, seconds):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "soak", script), str(seconds), "4242"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "soak ok" in out.stdout
