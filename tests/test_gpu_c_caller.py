"""GPU: a C program compiled against include/pll_amd.h and linked with libpll_amd.so - the drop-in
boundary exercised the way a C application would, not through ctypes."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_caller_reproduces_the_reference_kat(tmp_path):
    exe = str(tmp_path / "dropin")
    libdir = os.path.join(ROOT, "libpll-2_amd", "csrc")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_caller", "dropin.c"),
                           "-L" + libdir, "-lpll_amd", "-lm", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
    assert abs(float(lines["lnl"].split()[0]) - (-58.887310)) < 5.1e-7
