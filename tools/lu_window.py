import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
qcol = next(c for c in ("queue_id", "queue", "stream_id") if c in cols)
rows = cur.execute(f"select start, end, name, {qcol}, grid_x/workgroup_x from kernels order by start").fetchall()
fills = [i for i, r in enumerate(rows) if "fill_uniform" in r[2]]
rows = rows[fills[-1] + 1:]
t0 = rows[0][0]
lo, hi = float(sys.argv[2]) * 1e6, float(sys.argv[3]) * 1e6
for s, e, name, q, wg in rows:
    if lo <= s - t0 <= hi:
        short = name.replace("eigamd::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
        print(f"{(s - t0) / 1e3:10.1f} us +{(e - s) / 1e3:7.1f}  q{q} wg {wg:5d}  {short}")
