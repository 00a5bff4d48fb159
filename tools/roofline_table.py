#!/usr/bin/env python3
"""tools/roofline_table.py -- the per-kernel roofline table of DESIGN.md from the committed profile summaries.

usage: roofline_table.py <tag>          (reads profiles/<tag>_*_rocprof_summary.json, prints a markdown table)

Every figure is taken from the summary as tools/rocpd_summary.py wrote it: the walk kernel's average duration over the
traced launches (rocprofv3 --kernel-trace --stats), algorithmic bytes per launch / that duration, / 8 TB/s, and the
counter traffic (separate --pmc passes) over the algorithmic bytes.
"""
import glob
import json
import os
import sys

ORDER = ["c3", "c3_loadskip", "c3t", "c3u", "lds2", "c2", "c3_eager40", "c5", "c2_short", "c3_short", "c5_short", "c2_ragged", "c3_ragged", "c5_ragged"]
PEAK = 8000.0


def main():
    tag = sys.argv[1]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    rows = {}
    for f in glob.glob(os.path.join(root, tag + "_*_rocprof_summary.json")):
        wl = os.path.basename(f)[len(tag) + 1:-len("_rocprof_summary.json")]
        rows[wl] = json.load(open(f))
    print("| workload | kernel (rocprofv3 name) | launches traced | ms / launch | GB/s algorithmic | frac of 8 TB/s | traffic / algorithmic (PMC) |")
    print("|---|---|---|---|---|---|---|")
    for wl in ORDER + sorted(set(rows) - set(ORDER)):
        if wl not in rows:
            continue
        d = rows[wl]
        k = d["walk_kernel"]
        name = k["name"].replace("void fsmhip::", "").replace("fsmhip::", "").replace("(WalkArgs)", "")
        alg = d["pmc"].get("algorithmic_bytes_per_launch")
        gbps = alg / k["avg_ms"] / 1e6 if alg else k.get("algorithmic_GBps")
        tr = d["pmc"].get("traffic_over_algorithmic")
        note = ""
        if wl == "c3_loadskip":
            note = " (matched; %.2f of the bytes touched)" % (d["pmc"]["hbm_bytes_per_launch"] / alg) if alg else ""
        print("| %s | `%s` | %d | %.3f | %.0f%s | %s | %s |" % (
            wl, name, k["calls"], k["avg_ms"], gbps, note,
            "" if wl == "c3_loadskip" else "%.3f" % (gbps / PEAK), "" if tr is None else "%.3f" % tr))


if __name__ == "__main__":
    main()
