export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_odd
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 120 rocprofv3 --kernel-trace -d $OUT -o t -- ./build/gemm_check f64 N N 4000 4000 4000 2 0 1 0 1 0 > $OUT/log.txt 2>&1 || echo "(profiler exit $?)"
python3 - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/prof_odd/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select start, end, name, grid_x/workgroup_x, queue_id, stream_id from kernels where name like '%gemm_kernel%' order by start").fetchall()
t0 = rows[0][0]
for s, e, n, g, q, st in rows: print("%8.3f %8.3f ms  wgs %4d queue %s stream %s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, g, q, st, "GUARD" if "true>" in n else "interior"))
PY
