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
for _sfx in "sdcz":
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
    lib.ref_opstar_ex.argtypes = [ctypes.c_int] * 5 + [ctypes.c_int64] * 3 + [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
    lib.ref_simd.restype = ctypes.c_char_p
    lib.ref_set_cache_sizes.argtypes = [ctypes.c_long] * 3
    for name in ("ref_blocked_lu_d", "ref_blocked_lu_s", "ref_blocked_lu_c", "ref_blocked_lu_z"):
        f = getattr(lib, name)
        f.restype = ctypes.c_long
        f.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    if hasattr(lib, "ref_getrs"):
        lib.ref_getrs.restype = None
        lib.ref_getrs.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
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


def oracle_product_ex(opa, opb, m, n, k, alpha, a, lda, b, ldb, beta, c, c_inc, ldc):
    """The forms no F77 routine expresses: one operand of real scalars beside a complex one
    (read off the dtypes), and / or a destination whose rows lie c_inc elements apart."""
    dt = c.dtype
    kind = DTYPES.index(dt.type)
    a_real = int(a.dtype.kind == "f" and dt.kind == "c")
    b_real = int(b.dtype.kind == "f" and dt.kind == "c")
    al, be = np.array([alpha], dtype=dt), np.array([beta], dtype=dt)
    f = getattr(oracle, "oracle_product_ex_" + SUFFIX[kind])
    f.restype = None
    f.argtypes = [ctypes.c_int] * 6 + [ctypes.c_long] * 3 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long,
                                                             ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p,
                                                             ctypes.c_void_p, ctypes.c_long, ctypes.c_long]
    f(a_real, b_real, int(opa in "TC"), int(opa in "CJ"), int(opb in "TC"), int(opb in "CJ"), m, n, k, _vp(al), _vp(a),
      lda, _vp(b), ldb, _vp(be), _vp(c), c_inc, ldc)
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


def run_netlib_chk3_trsm(kind, trsm_fn, params):
    """Restated TRSM rows of ?CHK3 over `trsm_fn` (a ctypes function with the ?trsm_ signature)."""
    real = ctypes.c_float if kind in (0, 2) else ctypes.c_double
    idim = (ctypes.c_int * len(params["idim"]))(*params["idim"])
    alf = (real * len(params["alf"]))(*params["alf"])
    per = 2 if params["cplx"] else 1
    rep = NetlibReport()
    f = getattr(oracle, "netlib_chk3_trsm_" + SUFFIX[kind])
    f(ctypes.cast(trsm_fn, ctypes.c_void_p), len(params["idim"]), idim, len(params["alf"]) // per, alf, ctypes.byref(rep))
    return rep


def run_netlib_chke_trsm(kind, trsm_fn):
    return oracle.netlib_chke_trsm(ctypes.cast(trsm_fn, ctypes.c_void_p), ctypes.c_char("SDCZ"[kind].encode()))


def oracle_trsm(kind):
    return getattr(oracle, "oracle_trsm_" + SUFFIX[kind])


def oracle_getrs(kind):
    return getattr(oracle, "oracle_getrs_" + SUFFIX[kind])


def call_f77_getrs(fn, trans, n, nrhs, a, lda, ipiv, b, ldb):
    """Call any function with the F77 ?getrs_ signature on numpy buffers (b is overwritten); returns info."""
    ci = lambda v: ctypes.byref(ctypes.c_int(int(v)))
    info = ctypes.c_int(0)
    fn(ctypes.c_char_p(trans.encode()), ci(n), ci(nrhs), _vp(a), ci(lda), _vp(ipiv), _vp(b), ci(ldb), ctypes.byref(info))
    return info.value


def call_f77_trsm(fn, side, uplo, transa, diag, m, n, alpha, a, lda, b, ldb):
    """Call any function with the F77 ?trsm_ signature on numpy buffers (b is overwritten)."""
    al = np.array([alpha], dtype=b.dtype)
    ci = lambda v: ctypes.byref(ctypes.c_int(int(v)))
    cs = lambda c: ctypes.c_char_p(c.encode())
    fn(cs(side), cs(uplo), cs(transa), cs(diag), ci(m), ci(n), _vp(al), _vp(a), ci(lda), _vp(b), ci(ldb))
    return b


def trsm_operands(kind, side, uplo, diag, m, n, seed, pad=0, rogue=True):
    """A well-conditioned triangular A (m x m or n x n) and a general B (m x n), column-major
    with `pad` extra rows, from the library's counter-based generator.  Off-diagonal entries
    are U[-1,1) * 2/na, the diagonal 1.5 + 0.5 * U[-1,1); the triangle that ?trsm_ must not read
    (and the diagonal when diag = 'U') holds -1e10 when `rogue`."""
    import eigen_amd
    dt = DTYPES[kind]
    na = m if side == "L" else n
    a = eigen_amd.fill_uniform_host(dt, na, na, seed, ld=na + pad)
    b = eigen_amd.fill_uniform_host(dt, m, n, seed + 1, ld=m + pad)
    if na:
        i, j = np.indices((na, na))
        d = np.diagonal(a).copy()
        a *= 2.0 / na
        a[i == j] = 1.5 + 0.5 * d
        hide = (i > j) if uplo == "U" else (i < j)
        if diag == "U":
            hide = hide | (i == j)
        if rogue:
            a[hide] = -1.0e10
    return a, b


def dense_op_triangle(a, uplo, transa, diag):
    """op(A) as a dense matrix: the stored triangle, unit diagonal if asked, transposed /
    conjugated as `transa` says."""
    t = np.triu(a) if uplo == "U" else np.tril(a)
    if diag == "U":
        t = t - np.diag(np.diagonal(t)) + np.eye(a.shape[0], dtype=a.dtype)
    if transa == "T":
        t = t.T
    elif transa == "C":
        t = t.conj().T
    return t


def trsm_residual_ratio(kind, side, uplo, transa, diag, alpha, a, b0, x):
    """?MMCH-style componentwise ratio of op(A) X (or X op(A)) against alpha B:
    |op(A) X - alpha B| / (eps * (|op(A)| |X| + |alpha| |B|))."""
    abs1 = (lambda z: np.abs(z.real) + np.abs(z.imag)) if np.iscomplexobj(x) else np.abs
    t = dense_op_triangle(np.asarray(a), uplo, transa, diag).astype(np.complex128 if kind >= 2 else np.float64)
    xx, bb = x.astype(t.dtype), b0.astype(t.dtype)
    r = (t @ xx if side == "L" else xx @ t) - alpha * bb
    g = (abs1(t) @ abs1(xx) if side == "L" else abs1(xx) @ abs1(t)) + abs1(np.asarray(alpha)) * abs1(bb)
    g = np.where(g == 0, 1.0, g)
    return float((abs1(r) / (EPS[kind] * g)).max()) if x.size else 0.0


def run_netlib_chk4_syrk(kind, syrk_fn, params):
    """Restated SYRK rows of ?CHK4 over `syrk_fn` (a ctypes function with the ?syrk_ signature)."""
    real = ctypes.c_float if kind in (0, 2) else ctypes.c_double
    idim = (ctypes.c_int * len(params["idim"]))(*params["idim"])
    alf = (real * len(params["alf"]))(*params["alf"])
    bet = (real * len(params["bet"]))(*params["bet"])
    per = 2 if params["cplx"] else 1
    rep = NetlibReport()
    f = getattr(oracle, "netlib_chk4_syrk_" + SUFFIX[kind])
    f(ctypes.cast(syrk_fn, ctypes.c_void_p), len(params["idim"]), idim, len(params["alf"]) // per, alf,
      len(params["bet"]) // per, bet, ctypes.byref(rep))
    return rep


def run_netlib_chke_syrk(kind, syrk_fn):
    return oracle.netlib_chke_syrk(ctypes.cast(syrk_fn, ctypes.c_void_p), ctypes.c_char("SDCZ"[kind].encode()))


def oracle_syrk(kind):
    return getattr(oracle, "oracle_syrk_" + SUFFIX[kind])


def call_f77_syrk(fn, uplo, trans, n, k, alpha, a, lda, beta, c, ldc):
    """Call any function with the F77 ?syrk_ signature on numpy buffers (c is updated)."""
    al, be = np.array([alpha], dtype=c.dtype), np.array([beta], dtype=c.dtype)
    ci = lambda v: ctypes.byref(ctypes.c_int(int(v)))
    cs = lambda ch: ctypes.c_char_p(ch.encode())
    fn(cs(uplo), cs(trans), ci(n), ci(k), _vp(al), _vp(a), ci(lda), _vp(be), _vp(c), ci(ldc))
    return c


def syrk_operands(kind, uplo, trans, n, k, seed, pad=0):
    """A (n x k, or k x n for trans != 'N') and C (n x n) from the library's generator; the
    triangle of C that ?syrk_ must not touch holds -1e10."""
    import eigen_amd
    dt = DTYPES[kind]
    ar, ac = (n, k) if trans == "N" else (k, n)
    a = eigen_amd.fill_uniform_host(dt, ar, ac, seed, ld=ar + pad)
    c = eigen_amd.fill_uniform_host(dt, n, n, seed + 1, ld=n + pad)
    if n:
        i, j = np.indices((n, n))
        c[(i > j) if uplo == "U" else (i < j)] = -1.0e10
    return a, c


def run_netlib_chk4_herk(kind, herk_fn, params):
    """Restated HERK rows of ZCHK4 / CCHK4 over `herk_fn` (ctypes function, ?herk_ signature)."""
    real = ctypes.c_float if kind == 2 else ctypes.c_double
    idim = (ctypes.c_int * len(params["idim"]))(*params["idim"])
    alf = (real * len(params["alf"]))(*params["alf"])
    bet = (real * len(params["bet"]))(*params["bet"])
    rep = NetlibReport()
    f = getattr(oracle, "netlib_chk4_herk_" + SUFFIX[kind])
    f(ctypes.cast(herk_fn, ctypes.c_void_p), len(params["idim"]), idim, len(params["alf"]) // 2, alf,
      len(params["bet"]) // 2, bet, ctypes.byref(rep))
    return rep


def run_netlib_chke_herk(kind, herk_fn):
    return oracle.netlib_chke_herk(ctypes.cast(herk_fn, ctypes.c_void_p), ctypes.c_char("SDCZ"[kind].encode()))


def oracle_herk(kind):
    return getattr(oracle, "oracle_herk_" + SUFFIX[kind])


def call_f77_herk(fn, uplo, trans, n, k, alpha, a, lda, beta, c, ldc):
    """Call any function with the F77 ?herk_ signature on numpy buffers (alpha, beta real)."""
    rdt = np.float32 if c.dtype == np.complex64 else np.float64
    al, be = np.array([alpha], dtype=rdt), np.array([beta], dtype=rdt)
    ci = lambda v: ctypes.byref(ctypes.c_int(int(v)))
    cs = lambda ch: ctypes.c_char_p(ch.encode())
    fn(cs(uplo), cs(trans), ci(n), ci(k), _vp(al), _vp(a), ci(lda), _vp(be), _vp(c), ci(ldc))
    return c


def herk_operands(kind, uplo, trans, n, k, seed, pad=0):
    """A (n x k, or k x n for trans 'C') and C (n x n) from the library's generator; the triangle
    of C that ?herk_ must not touch holds -1e10, and so does the imaginary part of C's diagonal
    (it must be ignored, as ZMAKE 'HE' of the Netlib tester arranges)."""
    a, c = syrk_operands(kind, uplo, "N" if trans == "N" else "T", n, k, seed, pad)
    if n:
        i = np.arange(n)
        c[i, i] = c[i, i].real - 1.0e10j
    return a, c


def run_netlib_chk5(kind, fn, herm, params):
    """Restated ?CHK5 (SYR2K; HER2K when herm) over `fn` (ctypes function, ?syr2k_ / ?her2k_ signature)."""
    real = ctypes.c_float if kind in (0, 2) else ctypes.c_double
    idim = (ctypes.c_int * len(params["idim"]))(*params["idim"])
    alf = (real * len(params["alf"]))(*params["alf"])
    bet = (real * len(params["bet"]))(*params["bet"])
    per = 2 if params["cplx"] else 1
    rep = NetlibReport()
    f = getattr(oracle, "netlib_chk5_" + SUFFIX[kind])
    f(ctypes.cast(fn, ctypes.c_void_p), int(herm), len(params["idim"]), idim, len(params["alf"]) // per, alf,
      len(params["bet"]) // per, bet, ctypes.byref(rep))
    return rep


def run_netlib_chke_r2k(kind, fn, herm):
    return oracle.netlib_chke_r2k(ctypes.cast(fn, ctypes.c_void_p), ctypes.c_char("SDCZ"[kind].encode()), int(herm))


def oracle_r2k(kind, herm):
    return getattr(oracle, ("oracle_her2k_" if herm else "oracle_syr2k_") + SUFFIX[kind])


def call_f77_r2k(fn, herm, uplo, trans, n, k, alpha, a, lda, b, ldb, beta, c, ldc):
    """Call any function with the F77 ?syr2k_ / ?her2k_ signature (beta real for her2k)."""
    al = np.array([alpha], dtype=c.dtype)
    be = np.array([beta], dtype=(np.float32 if c.dtype == np.complex64 else np.float64) if herm else c.dtype)
    ci = lambda v: ctypes.byref(ctypes.c_int(int(v)))
    cs = lambda ch: ctypes.c_char_p(ch.encode())
    fn.restype = ctypes.c_int
    return fn(cs(uplo), cs(trans), ci(n), ci(k), _vp(al), _vp(a), ci(lda), _vp(b), ci(ldb), _vp(be), _vp(c), ci(ldc))


def r2k_operands(kind, herm, uplo, trans, n, k, seed, pad=0):
    """A, B (n x k, or k x n for trans != 'N') and C (n x n); the triangle of C that must not be
    touched holds -1e10, and for the Hermitian form so does the imaginary part of C's diagonal."""
    import eigen_amd
    dt = DTYPES[kind]
    ar, ac = (n, k) if trans == "N" else (k, n)
    a = eigen_amd.fill_uniform_host(dt, ar, ac, seed, ld=ar + pad)
    b = eigen_amd.fill_uniform_host(dt, ar, ac, seed + 7, ld=ar + pad)
    c = eigen_amd.fill_uniform_host(dt, n, n, seed + 1, ld=n + pad)
    if n:
        i, j = np.indices((n, n))
        c[(i > j) if uplo == "U" else (i < j)] = -1.0e10
        if herm:
            d = np.arange(n)
            c[d, d] = c[d, d].real - 1.0e10j
    return a, b, c


def run_netlib_chk2(kind, fn, herm, params):
    """Restated ?CHK2 (SYMM; HEMM when herm) over `fn` (ctypes function, ?symm_ signature)."""
    real = ctypes.c_float if kind in (0, 2) else ctypes.c_double
    idim = (ctypes.c_int * len(params["idim"]))(*params["idim"])
    alf = (real * len(params["alf"]))(*params["alf"])
    bet = (real * len(params["bet"]))(*params["bet"])
    per = 2 if params["cplx"] else 1
    rep = NetlibReport()
    f = getattr(oracle, "netlib_chk2_" + SUFFIX[kind])
    f(ctypes.cast(fn, ctypes.c_void_p), int(herm), len(params["idim"]), idim, len(params["alf"]) // per, alf,
      len(params["bet"]) // per, bet, ctypes.byref(rep))
    return rep


def run_netlib_chk3_trmm(kind, fn, params):
    """Restated TRMM rows of ?CHK3 over `fn` (ctypes function, ?trmm_ signature)."""
    real = ctypes.c_float if kind in (0, 2) else ctypes.c_double
    idim = (ctypes.c_int * len(params["idim"]))(*params["idim"])
    alf = (real * len(params["alf"]))(*params["alf"])
    per = 2 if params["cplx"] else 1
    rep = NetlibReport()
    f = getattr(oracle, "netlib_chk3_trmm_" + SUFFIX[kind])
    f(ctypes.cast(fn, ctypes.c_void_p), len(params["idim"]), idim, len(params["alf"]) // per, alf, ctypes.byref(rep))
    return rep


def call_f77_symm(fn, side, uplo, m, n, alpha, a, lda, b, ldb, beta, c, ldc):
    """Call any function with the F77 ?symm_ / ?hemm_ signature on numpy buffers (c is updated)."""
    al, be = np.array([alpha], dtype=c.dtype), np.array([beta], dtype=c.dtype)
    ci = lambda v: ctypes.byref(ctypes.c_int(int(v)))
    cs = lambda ch: ctypes.c_char_p(ch.encode())
    fn(cs(side), cs(uplo), ci(m), ci(n), _vp(al), _vp(a), ci(lda), _vp(b), ci(ldb), _vp(be), _vp(c), ci(ldc))
    return c


def symm_operands(kind, side, uplo, herm, m, n, seed, pad=0):
    """A (na x na, the uplo triangle meaningful, -1e10 in the other one and, for a Hermitian A,
    in the imaginary part of the diagonal), B and C (m x n)."""
    import eigen_amd
    dt = DTYPES[kind]
    na = m if side == "L" else n
    a = eigen_amd.fill_uniform_host(dt, na, na, seed, ld=na + pad)
    b = eigen_amd.fill_uniform_host(dt, m, n, seed + 1, ld=m + pad)
    c = eigen_amd.fill_uniform_host(dt, m, n, seed + 2, ld=m + pad)
    if na:
        i, j = np.indices((na, na))
        a[(i > j) if uplo == "U" else (i < j)] = -1.0e10
        if herm:
            a[i == j] = a[i == j].real - 1.0e10j
    return a, b, c


def dense_symmetric(a, uplo, herm):
    """The full symmetric / Hermitian matrix stored in one triangle of a."""
    t = np.triu(a) if uplo == "U" else np.tril(a)
    d = np.diag(np.diagonal(t))
    off = t - d
    if herm:
        return off + off.conj().T + d.real
    return off + off.T + d


def gauge_ratio(kind, got, ref, alpha, a_abs1, b_abs1, beta, c0, k):
    """Componentwise ?MMCH-style ratio |got - ref| / (eps * G) with
    G = abs1(alpha) * (|A| |B|)_ij + abs1(beta) * abs1(c0_ij)."""
    abs1 = (lambda z: np.abs(z.real) + np.abs(z.imag)) if np.iscomplexobj(got) else np.abs
    g = abs1(np.asarray(alpha)) * (a_abs1 @ b_abs1) + abs1(np.asarray(beta)) * abs1(c0)
    g = np.where(g == 0, 1.0, g)
    return float((abs1(got.astype(np.complex128) - ref.astype(np.complex128)) / (EPS[kind] * g)).max()) if got.size else 0.0
