"""The reference's OWN unit tests for the path -- test/product_large.cpp, product_small.cpp, product_mmtr.cpp,
product_trsolve.cpp, product_syrk.cpp, product_symm.cpp, product_trmm.cpp, product_selfadjoint.cpp and lu.cpp of the
Eigen tree -- compiled where they lie by
oracle/Makefile with -DEIGEN_USE_BLAS (the reference's EIGEN_TEST_EXTERNAL_BLAS configuration,
test/CMakeLists.txt:20-25) and linked against the HIP library, so every dense product, triangular
solve, rank update and blocked LU above Eigen's size threshold in them runs on the GPU.  Their
checks are Eigen's: isApprox with test_precision (test/main.h:411-417), 1e-3 float / 1e-6 double.
A test binary exits non-zero on the first failed VERIFY.  (test/product_extra.cpp is not in the
list: in this configuration its matrix-vector check at line 108 fails with the reference's own BLAS
alone -- ?gemv_ with a negative stride, a level-2 routine outside this library.)"""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["product_large", "product_small", "product_mmtr", "product_trsolve", "product_syrk", "product_symm", "product_trmm",
         "product_selfadjoint", "lu"]


@pytest.mark.parametrize("name", NAMES)
def test_reference_unit_test_on_the_hip_library(name):
    path = os.path.join(ROOT, "oracle", "_ref", "eigen_test_" + name)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (needs the reference tree at build time)")
    # r<N>: repetitions of every sub-test, s<N>: seed (test/main.h:764-813)
    out = subprocess.run([path, "r3", "s20241111"], capture_output=True, text=True, timeout=900)
    print(out.stdout[-3000:], out.stderr[-3000:])
    assert out.returncode == 0
    assert "Repeating each test 3 times" in out.stdout
