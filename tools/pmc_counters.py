#!/usr/bin/env python3
"""Per-kernel sums of every counter in a rocprofv3 --pmc rocpd database."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
for name, c, n, v in rows:
    print(f"{name.split('(')[0][-40:]:40s} {c:32s} launches {n:5d} sum {v:18.0f} per-launch {v/n:16.0f}")
