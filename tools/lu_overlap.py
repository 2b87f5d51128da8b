#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of the LU: how much of the panel kernels' time falls
inside the large trailing-update GEMM launches (look-ahead overlap), plus GPU idle time."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
t0, t1 = rows[0][1], max(r[2] for r in rows)
big = [(s, e) for n, s, e in rows if "gemm_kernel" in n and e - s > 500000]
small = [(s, e) for n, s, e in rows if "lu_" in n]
inside = 0
bi = 0
for s, e in small:
    while bi < len(big) and big[bi][1] < s: bi += 1
    if bi < len(big) and big[bi][0] <= s <= big[bi][1]: inside += 1
busy = 0; cur_end = t0
for n, s, e in rows:
    if s > cur_end: busy += 0
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
print(f"span {1e-6*(t1-t0):.1f} ms, union of kernel time {1e-6*busy:.1f} ms, big gemms {len(big)} total {1e-6*sum(e-s for s,e in big):.1f} ms, "
      f"panel kernels {len(small)} of which {inside} start inside a big gemm")
