#!/usr/bin/env python3
"""Where the time of the LU's base cases goes: factorises rows x cols panels (cols <= 16: one base-case launch each, no
rides) of several shapes; under `rocprofv3 --kernel-trace -d DIR -o NAME -- python3 tools/lu_panel_probe.py`,
`--read DB` prints the duration of every base-case launch beside its shape (launch order = the order below), from which
the cost per pivot column and the fixed part (load, store, launch) follow."""
import os, sys
SHAPES = [(rows, cols) for rows in (512, 1024, 2048, 4096, 8192, 16384) for cols in (4, 8, 16)]
KINDS = ("f64", "f32")
REPS = 3
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    import sqlite3
    cur = sqlite3.connect(sys.argv[2]).cursor()
    rows = cur.execute("select start, end, name, grid_x / workgroup_x from kernels where name like '%lu_base_%' order by start").fetchall()
    it = iter(rows)
    for kind in KINDS:
        for shape in SHAPES:
            best = None
            for _ in range(REPS):
                s, e, name, wg = next(it)
                best = min(best, e - s) if best else e - s
            short = name.replace("eigamd::", "").replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            print(f"{kind} {shape[0]:6d} x {shape[1]:2d}  {best / 1e3:7.1f} us  wg {wg:3d}  {short}")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eigen_amd
from eigen_amd.device import DeviceMatrix
from eigen_amd import lu as ealu
for kind in KINDS:
    for rows, cols in SHAPES:
        a = DeviceMatrix(rows, cols, {"f64": eigen_amd.F64, "f32": eigen_amd.F32}[kind])
        for _ in range(REPS):
            a.fill_uniform(77)
            ealu.getrf_dev(a)
torch.cuda.synchronize()
print("done", flush=True)
