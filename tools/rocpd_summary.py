#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace as a per-kernel stats table
(the same numbers `--stats` prints: calls, total, average, min, max, share)."""
import sqlite3
import sys


def main(db_path, out=sys.stdout):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {db_path}", file=out)
    print(f"{'kernel':60s} {'calls':>8s} {'total_ms':>12s} {'avg_us':>12s} {'min_us':>10s} {'max_us':>12s} {'pct':>7s}", file=out)
    for name, calls, tot, avg, mn, mx in rows:
        short = name.split("(")[0][-60:]
        print(f"{short:60s} {calls:8d} {tot/1e6:12.3f} {avg/1e3:12.2f} {mn/1e3:10.2f} {mx/1e3:12.2f} {100*tot/total:7.2f}", file=out)


if __name__ == "__main__":
    main(sys.argv[1])
