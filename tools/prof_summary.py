#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs into small text files for profiles/.

usage: prof_summary.py <prof_dir> <out_prefix>
  <prof_dir>/trace/*.db      from `rocprofv3 --kernel-trace --stats`
  <prof_dir>/pmc_*/*.db      from separate `rocprofv3 --pmc ...` passes
"""
import glob
import os
import sqlite3
import sys


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return con.execute(sql).fetchall()
    finally:
        con.close()


def main():
    d, out = sys.argv[1], sys.argv[2]
    lines = []
    for db in sorted(glob.glob(os.path.join(d, "trace", "*.db"))):
        lines.append(f"# kernel-trace stats ({os.path.relpath(db, d)}): name, calls, total_us, avg_us, pct")
        for name, calls, tot, avg, pct in q(db, "select name,total_calls,total_duration,average,percentage from top_kernels"):
            lines.append(f"{name[:90]:<90} {calls:>6} {tot / 1:>12.1f} {avg:>10.3f} {pct:>6.2f}")
        lines.append("# per-kernel resources: name, grid, workgroup, vgpr, sgpr, lds, scratch")
        for r in q(db, "select name, max(grid_x*grid_y*grid_z), max(workgroup_x), max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name"):
            lines.append("  " + " ".join(str(x)[:80] for x in r))
    for sub in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        for db in sorted(glob.glob(os.path.join(sub, "*.db"))):
            lines.append(f"# PMC ({os.path.relpath(db, d)}): kernel, counter, dispatches, mean value per dispatch")
            rows = q(db, "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name")
            for k, c, n, v in rows:
                if k.startswith("bs::") or k.startswith("void bs::"):
                    lines.append(f"{k[:70]:<70} {c:<22} {n:>5} {v:>16.2f}")
    open(out + ".txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
