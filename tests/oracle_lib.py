"""Loaders for the test-side checkers: the C oracle (oracle/_build/liboracle.so) and,
when present, the reference itself compiled under oracle/_ref/.  Imported by
conftest.py before eigen_amd so that the recording ``xerbla_`` of the oracle is
the one every library in the process calls."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")

DTYPES = [np.float32, np.float64, np.complex64, np.complex128]
SUFFIX = "sdcz"
EPS = [2.0 ** -23, 2.0 ** -52, 2.0 ** -23, 2.0 ** -52]  # machine epsilon per kind

if not os.path.exists(ORACLE_SO):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_build/liboracle.so"])
oracle = ctypes.CDLL(ORACLE_SO, mode=ctypes.RTLD_GLOBAL)
for _sfx in "sd":
    _f = getattr(oracle, "oracle_blocked_lu_" + _sfx)
    _f.restype = ctypes.c_long
    _f.argtypes = [ctypes.c_long, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_long]


class NetlibReport(ctypes.Structure):
    _fields_ = [("ncalls", ctypes.c_int), ("nfail_same", ctypes.c_int), ("nfail_ratio", ctypes.c_int),
                ("nfail_xerbla", ctypes.c_int), ("fatal", ctypes.c_int), ("errmax", ctypes.c_double)]


def _cpu_has(flag):
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return flag in line.split()
    except OSError:
        pass
    return False


def load_ref(variant=None):
    """The reference library built by oracle/Makefile, or None.  variant: 'novec',
    'v3', 'v4' or None for the widest ISA this host runs."""
    if variant is None:
        variant = "v4" if all(_cpu_has(f) for f in ("avx512f", "avx512dq", "avx512bw", "avx512vl", "avx512cd")) else "v3"
    path = os.path.join(REF_DIR, f"libeigen_ref_{variant}.so")
    if not os.path.exists(path):
        return None
    if variant == "v3" and not (_cpu_has("avx2") and _cpu_has("fma")):
        return None
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    lib.ref_bench_gemm.restype = ctypes.c_double
    lib.ref_bench_gemm.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
    lib.ref_bench_lu.restype = ctypes.c_double
    lib.ref_bench_lu.argtypes = [ctypes.c_int64, ctypes.c_int]
    lib.ref_opstar.argtypes = [ctypes.c_int] * 3 + [ctypes.c_int64] * 3 + [ctypes.c_void_p, ctypes.c_void_p,
                                                                        ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    lib.ref_simd.restype = ctypes.c_char_p
    lib.ref_set_cache_sizes.argtypes = [ctypes.c_long] * 3
    for name in ("ref_blocked_lu_d", "ref_blocked_lu_s"):
        f = getattr(lib, name)
        f.restype = ctypes.c_long
        f.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    lib.variant = variant
    return lib


def _vp(a):
    return ctypes.c_void_p(a.ctypes.data)


def call_f77(fn, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc):
    """Call any function with the F77 ?gemm_ signature on numpy buffers."""
    dt = c.dtype
    al, be = np.array([alpha], dtype=dt), np.array([beta], dtype=dt)
    ci = lambda v: ctypes.byref(ctypes.c_int(int(v)))
    fn(ctypes.c_char_p(transa.encode()), ctypes.c_char_p(transb.encode()), ci(m), ci(n), ci(k), _vp(al), _vp(a),
       ci(lda), _vp(b), ci(ldb), _vp(be), _vp(c), ci(ldc))
    return c


def oracle_gemm(kind):
    return getattr(oracle, "oracle_gemm_" + SUFFIX[kind])


def oracle_product64(kind, opa, opb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc):
    """general_matrix_matrix_product semantics incl. conjugate-without-transpose
    (op letters N, T, C, J)."""
    dt = c.dtype
    al, be = np.array([alpha], dtype=dt), np.array([beta], dtype=dt)
    f = getattr(oracle, "oracle_product64_" + SUFFIX[kind])
    f.argtypes = [ctypes.c_int] * 4 + [ctypes.c_long] * 3 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long,
                                                             ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p,
                                                             ctypes.c_void_p, ctypes.c_long]
    f(int(opa in "TC"), int(opa in "CJ"), int(opb in "TC"), int(opb in "CJ"), m, n, k, _vp(al), _vp(a), lda, _vp(b),
      ldb, _vp(be), _vp(c), ldc)
    return c


def last_xerbla():
    name = ctypes.create_string_buffer(8)
    info = oracle.netlib_last_xerbla(name)
    return info, name.value.decode()


def reset_xerbla():
    ctypes.c_int.in_dll(oracle, "netlib_xerbla_count").value = 0


def parse_blat3_dat(path):
    """Parameters of a Netlib level-3 tester input file (blas/testing/?blat3.dat):
    threshold, dimension list, alpha and beta lists, error-exit flag."""
    with open(path) as f:
        lines = [ln.strip() for ln in f.readlines()]

    def scalars(line, count):
        toks = line.replace("(", " ").replace(")", " ").replace(",", " ").split()
        vals = []
        for t in toks:
            try:
                vals.append(float(t))
            except ValueError:
                break
        return vals[:count]

    test_error_exits = lines[6].split()[0] == "T"
    thresh = float(lines[7].split()[0])
    nidim = int(lines[8].split()[0])
    idim = [int(t) for t in lines[9].split()[:nidim]]
    nalf = int(lines[10].split()[0])
    cplx = "(" in lines[11]
    alf = scalars(lines[11], nalf * (2 if cplx else 1))
    nbet = int(lines[12].split()[0])
    bet = scalars(lines[13], nbet * (2 if cplx else 1))
    return dict(test_error_exits=test_error_exits, thresh=thresh, idim=idim, alf=alf, bet=bet, cplx=cplx)


def run_netlib_chk1(kind, gemm_fn, params):
    """Restated ?CHK1 over `gemm_fn` (a ctypes function with the ?gemm_ signature)."""
    real = ctypes.c_float if kind in (0, 2) else ctypes.c_double
    idim = (ctypes.c_int * len(params["idim"]))(*params["idim"])
    alf = (real * len(params["alf"]))(*params["alf"])
    bet = (real * len(params["bet"]))(*params["bet"])
    per = 2 if params["cplx"] else 1
    rep = NetlibReport()
    f = getattr(oracle, "netlib_chk1_" + SUFFIX[kind])
    f(ctypes.cast(gemm_fn, ctypes.c_void_p), len(params["idim"]), idim, len(params["alf"]) // per, alf,
      len(params["bet"]) // per, bet, ctypes.byref(rep))
    return rep


def run_netlib_chke(kind, gemm_fn):
    return oracle.netlib_chke_gemm(ctypes.cast(gemm_fn, ctypes.c_void_p), ctypes.c_char("SDCZ"[kind].encode()))


def gauge_ratio(kind, got, ref, alpha, a_abs1, b_abs1, beta, c0, k):
    """Componentwise ?MMCH-style ratio |got - ref| / (eps * G) with
    G = abs1(alpha) * (|A| |B|)_ij + abs1(beta) * abs1(c0_ij)."""
    abs1 = (lambda z: np.abs(z.real) + np.abs(z.imag)) if np.iscomplexobj(got) else np.abs
    g = abs1(np.asarray(alpha)) * (a_abs1 @ b_abs1) + abs1(np.asarray(beta)) * abs1(c0)
    g = np.where(g == 0, 1.0, g)
    return float((abs1(got.astype(np.complex128) - ref.astype(np.complex128)) / (EPS[kind] * g)).max()) if got.size else 0.0
