#!/usr/bin/env python3
"""The device LU of one matrix many times over: every run must give the bits of the first (transpositions and factors).
   lu_repeat.py [n=8192] [runs=30] [kind=f64|f32|c64|c32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eigen_amd
from eigen_amd.device import DeviceMatrix
from eigen_amd import lu as ealu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
kind = {"f32": eigen_amd.F32, "f64": eigen_amd.F64, "c32": eigen_amd.C32, "c64": eigen_amd.C64}[sys.argv[3] if len(sys.argv) > 3 else "f64"]
a = DeviceMatrix(n, n, kind)
first = None
bad = 0
for r in range(runs):
    a.fill_uniform(77)
    transp, nb, fz = ealu.getrf_dev(a)
    torch.cuda.synchronize()
    cur = (transp.clone(), torch.view_as_real(a.as_strided()).clone() if kind >= 2 else a.as_strided().clone(), nb, fz)
    if first is None:
        first = cur
    else:
        same = torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1]) and cur[2:] == first[2:]
        if not same:
            bad += 1
            print(f"run {r}: differs from the first run (transpositions equal: {torch.equal(cur[0], first[0])})", flush=True)
print(f"LU {sys.argv[3] if len(sys.argv) > 3 else 'f64'} n={n}: {runs} runs, {bad} differ from the first", flush=True)
sys.exit(1 if bad else 0)
