#!/bin/bash
# Kernel trace of the device LU at one size: per-kernel totals and the busy fraction.
set -o pipefail
export TMPDIR=/tmp
n=${1:-16384}
OUT=$PWD/gpurun_out/prof_lu
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o lu -- python3 tools/lu_time.py $n > $OUT/trace.log 2>&1 || echo "(profiler exit status $?)"
grep "LU f64" $OUT/trace.log
python3 tools/rocpd_summary.py $(find $OUT/trace -name "*.db") | tee $OUT/summary.txt
python3 - $(find $OUT/trace -name "*.db") <<'PY' | tee -a $OUT/summary.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select start, end, name from kernels order by start").fetchall()
fills = [i for i, r in enumerate(rows) if "fill_uniform" in r[2]]
rows = rows[fills[-1] + 1:]          # second factorisation only
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_end = 0, t0
for s, e, _ in rows:
    if e <= cur_end: continue
    busy += e - max(s, cur_end); cur_end = e
print(f"second factorisation: kernels {len(rows)}  span {(t1-t0)/1e6:.1f} ms  union busy {busy/1e6:.1f} ms ({100*busy/(t1-t0):.0f}%)")
cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
qcol = next((c for c in ("queue_id", "queue", "stream_id") if c in cols), None)
if qcol:
    per = {}
    for s, e, name, qid in cur.execute(f"select start, end, name, {qcol} from kernels where start >= ? order by start", (t0,)):
        short = name.replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1].split("<")[0]
        if "gemm_kernel" in name: short = "gemm_kernel (>0.3 ms)" if e - s > 300000 else "gemm_kernel (small)"
        d = per.setdefault(qid, {})
        c = d.setdefault(short, [0, 0]); c[0] += 1; c[1] += e - s
    for qid, d in sorted(per.items()):
        tot = sum(v[1] for v in d.values())
        parts = ", ".join(f"{k} {v[0]} x = {v[1]/1e6:.1f}" for k, v in sorted(d.items(), key=lambda kv: -kv[1][1]))
        print(f"  queue {qid} kernel time {tot/1e6:.1f} ms: {parts}")
q = "select grid_x/workgroup_x as b, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like '%lu_base_%' group by b order by b"
for r in cur.execute(q).fetchall()[::16]: print("lu_base_fused workgroups %4d calls %4d avg %7.1f us min %7.1f us" % r)
PY
