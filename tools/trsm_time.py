#!/usr/bin/env python3
"""Times eigamd_trsm_dev (n x n triangle, n right-hand sides, n^3 flop) for a few forms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eigen_amd
from eigen_amd.device import DeviceMatrix, trsm_dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for kind, name in ((eigen_amd.F64, "d"), (eigen_amd.F32, "s"), (eigen_amd.C64, "z")):
    nn = n if kind != eigen_amd.C64 else n // 2
    a = DeviceMatrix(nn, nn, kind).fill_uniform(5)
    a.as_strided().mul_(2.0 / nn)
    a.as_strided().diagonal().fill_(1.5)
    b = DeviceMatrix(nn, nn, kind)
    for form in ("LLNN", "LUTN", "RLNN", "RUNU"):
        best = 1e30
        for _ in range(3):
            b.fill_uniform(6)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            trsm_dev(form[0], form[1], form[2], form[3], 1.0, a, b)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        flops = (4.0 if kind == eigen_amd.C64 else 1.0) * float(nn) ** 3
        print(f"{name}trsm {form} n={nn}: {best:.2f} ms  {flops / best / 1e9:.2f} TFLOP/s", flush=True)
