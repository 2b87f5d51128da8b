#!/usr/bin/env python3
"""Per-queue timeline of the device LU from a rocprofv3 kernel trace (rocpd SQLite): kernel time,
short gaps between consecutive kernels of a queue (launch latency), long gaps (the queue waits
for the other one), in windows of the factorisation.
usage: lu_timeline.py <results.db> [window_ms]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 50e6
cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
qcol = next(c for c in ("queue_id", "queue", "stream_id") if c in cols)
rows = cur.execute(f"select start, end, name, {qcol} from kernels order by start").fetchall()
fills = [i for i, r in enumerate(rows) if "fill_uniform" in r[2]]
rows = rows[fills[-1] + 1:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
print(f"span {(t1 - t0) / 1e6:.1f} ms")
queues = sorted({r[3] for r in rows})
nwin = int((t1 - t0) // win) + 1
for q in queues:
    ks = [r for r in rows if r[3] == q]
    if len(ks) < 10:
        continue
    busy = [0.0] * nwin
    short = [0.0] * nwin
    long_ = [0.0] * nwin
    nshort = [0] * nwin
    prev_end = None
    for s, e, name, _ in ks:
        w = int((s - t0) // win)
        busy[w] += e - s
        if prev_end is not None and s > prev_end:
            g = s - prev_end
            if g < 20e3:
                short[w] += g
                nshort[w] += 1
            else:
                long_[w] += g
        prev_end = max(prev_end or e, e)
    print(f"queue {q}: {len(ks)} kernels, busy {sum(busy) / 1e6:.1f} ms, short gaps {sum(short) / 1e6:.1f} ms ({sum(nshort)}), "
          f"long gaps {sum(long_) / 1e6:.1f} ms")
    for w in range(nwin):
        print(f"   [{w * win / 1e6:5.0f} ms) busy {busy[w] / 1e6:6.1f}  short gaps {short[w] / 1e6:5.1f} ({nshort[w]:4d})  waits {long_[w] / 1e6:6.1f}")
