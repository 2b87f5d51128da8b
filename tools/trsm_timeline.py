#!/usr/bin/env python3
"""One triangular solve (eigamd_trsm_dev, double, left / lower / no transpose / non-unit, n x n with n right-hand sides) run
three times; under `rocprofv3 --kernel-trace -d DIR -o NAME -- python3 tools/trsm_timeline.py n`, `--read DB` then prints the
launches of the last run in order: start offset, duration, gap to the previous launch's end."""
import os, sys
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    import sqlite3
    cur = sqlite3.connect(sys.argv[2]).cursor()
    rows = cur.execute("select start, end, name, grid_x / workgroup_x from kernels order by start").fetchall()
    fills = [i for i, r in enumerate(rows) if "fill_uniform" in r[2]]
    rows = rows[fills[-1] + 1:]
    t0, prev = rows[0][0], rows[0][0]
    tot = 0
    for s, e, name, wg in rows:
        short = name.replace("eigamd::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev) / 1e3:6.1f}  wg {wg:5d}  {short}")
        prev = max(prev, e)
        tot += e - s
    print(f"span {(prev - t0) / 1e3:.1f} us, kernel time {tot / 1e3:.1f} us, launches {len(rows)}")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eigen_amd
from eigen_amd.device import DeviceMatrix, trsm_dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a = DeviceMatrix(n, n, eigen_amd.F64).fill_uniform(5)
a.as_strided().mul_(2.0 / n)
a.as_strided().diagonal().fill_(1.5)
b = DeviceMatrix(n, n, eigen_amd.F64)
for _ in range(3):
    torch.cuda.synchronize()
    b.fill_uniform(6)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    trsm_dev("L", "L", "N", "N", 1.0, a, b)
    e1.record()
    torch.cuda.synchronize()
    print(f"dtrsm LLNN n={n}: {e0.elapsed_time(e1):.3f} ms  {float(n) ** 3 / e0.elapsed_time(e1) / 1e9:.2f} TFLOP/s", flush=True)
