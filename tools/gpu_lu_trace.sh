#!/bin/bash
# Kernel trace of the device LU at one size: per-kernel totals and the busy fraction.
set -o pipefail
export TMPDIR=/tmp
n=${1:-16384}
OUT=$PWD/gpurun_out/prof_lu
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o lu -- python3 tools/lu_time.py $n > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
tail -2 $OUT/trace.log
python3 tools/rocpd_summary.py $(find $OUT/trace -name "*.db") | tee $OUT/summary.txt
python3 - $(find $OUT/trace -name "*.db") <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select start, end, name from kernels order by start").fetchall()
# second factorisation only (after the second fill kernel)
fills = [i for i, r in enumerate(rows) if "fill_uniform" in r[2]]
rows = rows[fills[-1] + 1:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_end = 0, t0
for s, e, _ in rows:
    if e <= cur_end: continue
    busy += e - max(s, cur_end); cur_end = e
print(f"kernels {len(rows)}  span {(t1-t0)/1e6:.1f} ms  union busy {busy/1e6:.1f} ms ({100*busy/(t1-t0):.0f}%)")
by = {}
for s, e, nme in rows:
    k = nme.split("(")[0][-60:]
    d = by.setdefault(k, [0, 0]); d[0] += 1; d[1] += e - s
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"{t/1e6:9.2f} ms {c:7d} calls {t/c/1e3:9.1f} us avg  {k}")
PY
