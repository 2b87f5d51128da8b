"""Pins the LU oracle (oracle/lu_oracle.c) against the reference's own blocked_lu:
identical pivots and zero-pivot bookkeeping, factors equal to rounding (bitwise for
the unblocked base case against the scalar build; for the complex kinds the column scaling
l = a / pivot is the textbook quotient in the oracle (-fcx-limited-range) and the scaled one of
std::complex in the reference, so those factors agree to rounding only)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol

SHAPES = [(1, 1), (5, 5), (16, 16), (17, 17), (40, 40), (100, 100), (129, 129), (300, 300), (700, 700), (300, 200),
          (200, 300), (10, 40)]


SFX = {np.float32: "s", np.float64: "d", np.complex64: "c", np.complex128: "z"}


def random_matrix(rng, shape, dt):
    a = rng.uniform(-1, 1, shape)
    if np.issubdtype(dt, np.complexfloating):
        a = a + 1j * rng.uniform(-1, 1, shape)
    return np.asfortranarray(a.astype(dt))


def oracle_lu(a, max_block=256):
    sfx = SFX[a.dtype.type]
    m, n = a.shape
    t = np.zeros(max(min(m, n), 1), dtype=np.int32)
    nb = ctypes.c_int()
    ld = a.strides[1] // a.itemsize if n > 1 else m
    r = getattr(ol.oracle, "oracle_blocked_lu_" + sfx)(m, n, a.ctypes.data, max(ld, 1), t.ctypes.data, ctypes.byref(nb),
                                                       max_block)
    return t[:min(m, n)], nb.value, r


def ref_lu(ref, a):
    sfx = SFX[a.dtype.type]
    m, n = a.shape
    t = np.zeros(max(min(m, n), 1), dtype=np.int32)
    nb = ctypes.c_int()
    ld = a.strides[1] // a.itemsize if n > 1 else m
    r = getattr(ref, "ref_blocked_lu_" + sfx)(m, n, a.ctypes.data, max(ld, 1), t.ctypes.data, ctypes.byref(nb))
    return t[:min(m, n)], nb.value, r


def residual(a0, lu, transp):
    """|| P*A - L*U ||_F / || A ||_F on the leading min(m,n) columns the blocked algorithm covers."""
    m, n = a0.shape
    size = min(m, n)
    wide = np.complex128 if np.iscomplexobj(a0) else np.float64
    pa = a0.astype(wide).copy()
    for i, p in enumerate(transp):
        if p != i:
            pa[[i, p], :] = pa[[p, i], :]
    l = np.tril(lu[:, :size].astype(wide), -1) + np.eye(m, size)
    u = np.triu(lu[:size, :].astype(wide))
    cols = n if m <= 16 else size          # the reference's blocked branch never touches columns >= size
    return np.linalg.norm((pa - l @ u)[:, :cols]) / np.linalg.norm(a0)


@pytest.mark.parametrize("variant", ["novec", "v3"])
@pytest.mark.parametrize("dt", [np.float64, np.float32, np.complex128, np.complex64])
def test_lu_oracle_vs_reference(variant, dt):
    ref = ol.load_ref(variant)
    if ref is None:
        pytest.skip("oracle/_ref not built")
    if variant == "novec":
        ref.ref_set_cache_sizes(32 * 1024, 256 * 1024, 2 * 1024 * 1024)
    eps = np.finfo(dt).eps
    for (m, n) in SHAPES:
        rng = np.random.default_rng(m * 1000 + n)
        a = random_matrix(rng, (m + 3, n), dt)
        a1, a2 = a.copy(order="F"), a.copy(order="F")
        t1, nb1, r1 = oracle_lu(a1[:m])
        t2, nb2, r2 = ref_lu(ref, a2[:m])
        assert np.array_equal(t1, t2) and nb1 == nb2 and r1 == r2 == -1
        assert np.array_equal(a1[m:], a[m:])  # padding rows untouched
        if min(m, n) <= 16 and variant == "novec" and not np.iscomplexobj(a):
            assert np.array_equal(a1, a2)
        else:
            assert np.abs(a1[:m] - a2[:m]).max() <= 64 * eps * max(m, n) * np.abs(a2[:m]).max()
        assert residual(a[:m], a1[:m], t1) <= 8 * eps * max(m, n)


def test_lu_oracle_zero_pivot():
    rng = np.random.default_rng(3)
    a = np.asfortranarray(rng.uniform(-1, 1, (40, 40)))
    a[:, 7] = 0.0
    a1 = a.copy(order="F")
    t, nb, r = oracle_lu(a1)
    assert r == 7
    ref = ol.load_ref()
    if ref is not None:
        a2 = a.copy(order="F")
        t2, nb2, r2 = ref_lu(ref, a2)
        assert r2 == 7 and np.array_equal(t, t2)
