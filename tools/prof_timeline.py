#!/usr/bin/env python3
"""Timeline of one steady-state batch from a rocprofv3 --kernel-trace run (rocpd sqlite):
start offset, duration and queue of every kernel between two consecutive k_prepass launches.

usage: prof_timeline.py <trace.db> [batch_index_from_end=3]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
    starts = [i for i, r in enumerate(rows) if "k_prepass" in r[0]]
    if len(starts) < back + 1:
        print("not enough batches")
        return
    a, b = starts[-back - 1], starts[-back]
    t0 = rows[a][1]
    # kernels of later batches can start before the next k_prepass only on the side streams: include by time window
    t1 = rows[b][1]
    print(f"# batch window {(t1 - t0) / 1000:.1f} us (k_prepass to next k_prepass); columns: start_us dur_us queue name")
    for name, s, e, qid, sid in rows:
        if t0 <= s < t1:
            short = name.split("(")[0].replace("void bs::", "").replace("bs::", "")
            print(f"{(s - t0) / 1000:8.2f} {(e - s) / 1000:8.2f}  q{qid} {short[:60]}")


if __name__ == "__main__":
    main()
