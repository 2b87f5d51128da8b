#!/usr/bin/env python3
"""Summarises rocprofv3's rocpd SQLite output (ROCm 7.2 default format) as text:
per-kernel time statistics (what `--stats` reports) and, for --pmc runs, the
per-dispatch counter values aggregated per kernel.
usage: rocpd_summary.py <results.db> [...]"""
import sqlite3
import sys


def short(name, n=110):
    return name if len(name) <= n else name[:n - 3] + "..."


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print(f"== {path}")
        rows = cur.execute(
            "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
            "from kernels group by name order by 3 desc").fetchall()
        total = sum(r[2] for r in rows) or 1
        print(f"{'calls':>6} {'total_ms':>12} {'avg_ms':>12} {'min_ms':>12} {'max_ms':>12} {'pct':>6}  kernel")
        for name, calls, tot, avg, mn, mx in rows:
            print(f"{calls:6d} {tot / 1e6:12.4f} {avg / 1e6:12.4f} {mn / 1e6:12.4f} {mx / 1e6:12.4f} "
                  f"{100.0 * tot / total:6.2f}  {short(name)}")
        try:
            cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
            if cols:
                rows = cur.execute(
                    "select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                    "from counters_collection group by kernel_name, counter_name order by 1, 2").fetchall()
                if rows:
                    print(f"-- counters (per dispatch): {'n':>4} {'avg':>18} {'min':>18} {'max':>18}")
                    for k, c, n, avg, mn, mx in rows:
                        print(f"   {c:32s} {n:4d} {avg:18.1f} {mn:18.1f} {mx:18.1f}  {short(k, 70)}")
        except sqlite3.Error as e:
            print("   (no counter view:", e, ")")
        db.close()


if __name__ == "__main__":
    main()
