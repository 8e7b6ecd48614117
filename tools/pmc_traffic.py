#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite).
gfx950 corrections (MI355X_MICROARCH.md, HBM section; re-calibrated here for 8-byte-per-lane loads with
tools/probes/stream8.hip: 1 GiB read -> FETCH_SIZE 0.5000 GiB, 0.25 GiB written -> WRITE_SIZE 0.2500 GiB):
FETCH_SIZE (KB) is doubled, WRITE_SIZE (KB) is taken as is."""
import json
import sqlite3
import sys


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    out = {}
    for name, n, tot in db.execute("select kernel_name, count(*), sum(value) from counters_collection "
                                   "where counter_name=? group by kernel_name", (counter,)):
        out[name.split("(")[0]] = (n, tot * 1024.0)
    return out


def main(fetch_db, write_db):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    res = {}
    print(f"{'kernel':45s} {'launches':>9s} {'read GB':>10s} {'written GB':>11s} {'bytes/launch':>14s}")
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0))[1] * 2 + w.get(k, (0, 0))[1])):
        n = f.get(k, w.get(k))[0]
        rd = 2.0 * f.get(k, (0, 0.0))[1]
        wr = w.get(k, (0, 0.0))[1]
        res[k] = dict(launches=n, read_bytes=rd, written_bytes=wr, bytes_per_launch=(rd + wr) / max(n, 1))
        print(f"{k[-45:]:45s} {n:9d} {rd/1e9:10.3f} {wr/1e9:11.3f} {(rd+wr)/max(n,1):14.0f}")
    return res


if __name__ == "__main__":
    r = main(sys.argv[1], sys.argv[2])
    if len(sys.argv) > 3:
        json.dump(r, open(sys.argv[3], "w"), indent=1)
