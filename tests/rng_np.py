"""numpy restatement of the library's counter-based U[-1,1) generator
(eigen_amd/csrc/rng.hpp) for arbitrary index sets: lets a test regenerate a few
rows or columns of a 16384 x 16384 operand without materialising it."""
import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z):
    with np.errstate(over="ignore"):
        z = z + _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def entries(dtype, seed, i, j):
    """Elements (i, j) (broadcast index arrays) of the matrix seeded with `seed`."""
    dtype = np.dtype(dtype)
    i = np.asarray(i, dtype=np.uint64)
    j = np.asarray(j, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _mix(np.uint64(seed) ^ _mix((j << np.uint64(32)) | i))
        h2 = _mix(h + _G)

    def f64(x):
        return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 4503599627370496.0) - 1.0

    def f32(x):
        return ((x >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 8388608.0) - np.float32(1.0))

    if dtype == np.float32:
        return f32(h)
    if dtype == np.float64:
        return f64(h)
    if dtype == np.complex64:
        return (f32(h) + 1j * f32(h2)).astype(np.complex64)
    return f64(h) + 1j * f64(h2)


def matrix(dtype, seed, rows, cols):
    return entries(dtype, seed, np.arange(rows)[:, None], np.arange(cols)[None, :])
