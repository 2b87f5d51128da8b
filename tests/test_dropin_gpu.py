"""Drop-in check: ordinary Eigen programs (tests/dropin/dropin_check.cpp, compiled
against the reference headers by oracle/Makefile) whose products and PartialPivLU run
on the HIP library -- once through the reference's own EIGEN_USE_BLAS binding and
once through include/eigen_amd/EigenSeam.h.  Each product is compared inside the
program with Eigen's coefficient-based lazyProduct."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("binary", ["dropin_blas", "dropin_seam"])
def test_dropin_program(binary):
    path = os.path.join(ROOT, "oracle", "_ref", binary)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (needs the reference tree at build time)")
    out = subprocess.run([path], capture_output=True, text=True, timeout=300)
    print(out.stdout[-3000:], out.stderr[-2000:])
    assert out.returncode == 0
    assert "DROPIN PASS" in out.stdout
    assert out.stdout.count("PASS ") >= 34 and "FAIL " not in out.stdout
