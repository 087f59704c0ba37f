"""Text summary of rocprofv3 sqlite outputs (ROCm 7.2 writes *_results.db): per-kernel launch count, mean / median / max duration from a
--kernel-trace run, and per-kernel means of the counters of any --pmc runs.
usage: python tools/prof_db_summary.py <dir with *.db files (searched recursively)> [kernel-substring ...]"""
import glob
import os
import sqlite3
import sys

import numpy as np


def main():
    root = sys.argv[1]
    pats = sys.argv[2:]
    want = lambda n: not pats or any(p in n for p in pats)
    short = lambda n: n.split("(")[0][:90]
    for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        tables = {r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")}
        rel = os.path.relpath(db, root)
        if "kernels" in tables:
            rows = con.execute("select name, duration, grid_x, workgroup_x from kernels").fetchall()
            per = {}
            for name, dur, grid, wg in rows:
                if want(name):
                    per.setdefault(short(name), []).append((dur / 1e3, grid, wg))
            if per:
                print(f"== {rel}: kernel trace (durations in us)")
                print(f"{'kernel':92s} {'calls':>6s} {'mean':>10s} {'median':>10s} {'max':>10s} {'grid':>8s} {'wg':>5s}")
                for k, v in sorted(per.items(), key=lambda kv: -sum(d for d, _, _ in kv[1])):
                    d = np.array([x[0] for x in v])
                    print(f"{k:92s} {len(v):6d} {d.mean():10.2f} {np.median(d):10.2f} {d.max():10.2f} {int(np.median([x[1] for x in v])):8d} {int(np.median([x[2] for x in v])):5d}")
        if "counters_collection" in tables:
            rows = con.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
            per = {}
            for name, cname, v in rows:
                if want(name):
                    per.setdefault((short(name), cname), []).append(float(v))
            if per:
                print(f"== {rel}: counters (mean per launch)")
                for (k, cname), v in sorted(per.items()):
                    print(f"{k:92s} {cname:22s} {np.mean(v):16.1f}  ({len(v)} launches)")
        con.close()


if __name__ == "__main__":
    main()
