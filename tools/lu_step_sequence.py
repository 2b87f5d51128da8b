#!/usr/bin/env python3
"""Kernel by kernel through one outer step of the device LU, from a rocprofv3 kernel trace (rocpd SQLite):
start time, queue, kernel, duration, gap to the queue's previous kernel, workgroups.
usage: lu_step_sequence.py <results.db> [fraction of the span at which to start, default 0.8] [kernels to print]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
qcol = next(c for c in ("queue_id", "queue", "stream_id") if c in cols)
rows = cur.execute(f"select start, end, name, {qcol}, grid_x, workgroup_x from kernels order by start").fetchall()
fills = [i for i, r in enumerate(rows) if "fill_uniform" in r[2]]
rows = rows[fills[-1] + 1:]
t0, t1 = rows[0][0], max(r[1] for r in rows)


def short(n):
    return n.replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1].split("<")[0]


i0 = next(i for i, r in enumerate(rows) if r[0] > t0 + frac * (t1 - t0) and "swap_plan" in r[2])
prev_end = {}
for r in rows[max(i0 - 4, 0): i0 + count]:
    q = r[3]
    gap = (r[0] - prev_end.get(q, r[0])) / 1e3
    prev_end[q] = r[1]
    print(f"{(r[0] - t0) / 1e6:9.3f} ms q{q} {short(r[2]):24s} {(r[1] - r[0]) / 1e3:7.1f} us  gap {gap:6.1f}  workgroups {r[4] // max(r[5], 1)}")
