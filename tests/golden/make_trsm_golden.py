#!/usr/bin/env python3
"""Generates tests/golden/trsm_golden.npz from the reference itself.

Run in the build container (needs oracle/_ref, i.e. /root/reference at build time):
    python tests/golden/make_trsm_golden.py

Inputs are NOT stored: tests regenerate them from (kind, side, uplo, diag, m, n, seed, pad)
with oracle_lib.trsm_operands (the library's counter-based generator plus a fixed recipe for
the triangle).  Stored per case: the parameters and the solution Eigen's F77 entry point
?trsm_ (blas/level3_impl.h:78-178) returned, from the AVX2+FMA build of the reference.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402


def case_list():
    cases = []
    sizes = [(1, 1), (2, 5), (7, 3), (33, 17), (64, 64), (129, 70), (130, 257), (300, 45)]
    idx = 0
    for kind in range(4):
        alphas = [1.0, 0.7, -1.3] if kind < 2 else [1.0 + 0j, 0.7 - 0.9j, -1.0 + 0.5j]
        for si, (m, n) in enumerate(sizes):
            for side in "LR":
                for uplo in "UL":
                    for trans in "NTC":
                        for diag in "NU":
                            idx += 1
                            if si >= 3 and idx % 3:       # larger shapes: a third of the forms each
                                continue
                            if kind < 2 and trans == "C" and si % 2:
                                continue                   # 'C' == 'T' for real kinds: keep a few
                            cases.append(dict(kind=kind, side=side, uplo=uplo, trans=trans, diag=diag, m=m, n=n,
                                              alpha=alphas[idx % 3], pad=idx % 3, seed=1000 + idx))
    return cases


def main():
    ref = ol.load_ref("v3")
    if ref is None:
        raise SystemExit("oracle/_ref/libeigen_ref_v3.so missing: run make -C oracle (needs /root/reference)")
    cases = case_list()
    out, meta = {}, []
    for i, c in enumerate(cases):
        a, b = ol.trsm_operands(c["kind"], c["side"], c["uplo"], c["diag"], c["m"], c["n"], c["seed"], c["pad"])
        fn = getattr(ref, ol.SUFFIX[c["kind"]] + "trsm_")
        na = c["m"] if c["side"] == "L" else c["n"]
        ol.call_f77_trsm(fn, c["side"], c["uplo"], c["trans"], c["diag"], c["m"], c["n"], c["alpha"], a, na + c["pad"], b,
                         c["m"] + c["pad"])
        x = np.array(b)
        stride = 1 if x.size <= 1500 else 5
        out[f"x{i}"] = x[::stride, ::stride].copy()
        cc = dict(c)
        al = complex(c["alpha"])
        cc["alpha"] = [al.real, al.imag]
        cc["stride"] = stride
        meta.append(cc)
    out["meta"] = np.frombuffer(json.dumps(dict(cases=meta)).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "trsm_golden.npz")
    np.savez_compressed(path, **out)
    print(f"{len(cases)} cases -> {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()
