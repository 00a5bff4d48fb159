#!/usr/bin/env python3
"""tools/rocpd_top_kernels.py -- print the per-kernel time table of a rocprofv3 --kernel-trace --stats run
(rocpd .db under <dir>): name, calls, total / average duration.  usage: rocpd_top_kernels.py <dir> [match]"""
import glob
import os
import sqlite3
import sys


def main():
    d = sys.argv[1]
    match = sys.argv[2] if len(sys.argv) > 2 else ""
    for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        con = sqlite3.connect(db)
        for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            if match in name:
                print(f"{calls:6d} calls  total {total / 1e3:10.3f} ms  avg {avg:10.1f} us  {pct:6.2f} %  {name[:150]}")   # the view is in microseconds
        con.close()


if __name__ == "__main__":
    main()
