#!/usr/bin/env python3
"""tools/rocpd_summary.py -- condense tools/profile.sh output (rocprofv3 rocpd .db files) into a
small JSON summary under profiles/, plus profiles/pmc_<workload>.json that bench.py reads for
`roofline.traffic`.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts 128-byte
requests as 64 bytes for 16-byte-per-lane coalesced streams (MI355X_MICROARCH.md section HBM), so the
read side is doubled; WRITE_SIZE is taken as reported (uncalibrated).
usage: rocpd_summary.py <prof_dir> <workload> <out_prefix>
"""
import glob
import json
import os
import sqlite3
import sys


def rows(db, sql, params=()):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cur.execute(sql, params)
    cols = [d[0] for d in cur.description]
    return [dict(zip(cols, r)) for r in cur.fetchall()]


def lines_summary(prof, wl, out, bench, top, walk, pmc, fetch_kib, write_kib, req=None):
    """The packed-lines front (bench.py --workload c3_short ...): algorithmic bytes = sum(len) + 8 B offsets + 4 B result per line.
    The lines are read by per-lane 16-byte buffer loads at 36-byte (short) or 512-byte (ragged) mean pitch, not by the
    16-byte-per-lane coalesced stream the x2 correction of FETCH_SIZE was calibrated on: the raw counter is taken at face value
    when it already covers >= 0.75 of the bytes that must be read, doubled otherwise -- both figures are recorded."""
    n = bench["config"]["lines"]
    alg = bench["roofline"]["algorithmic_bytes_per_launch"]
    must_read = bench["config"]["line_bytes"] + 8 * n
    raw = fetch_kib * 1024
    read = raw if raw >= 0.75 * must_read else 2 * raw
    w_, kind = wl.split("_")
    calib = None
    if kind == "ragged":
        # walk_ragged fetches 128-byte segments that start at any byte: a segment spans two memory lines and the counter tallies such
        # requests at anything between half and all of their bytes.  Calibrated, as MI355X_MICROARCH.md (HBM) asks for access patterns
        # other than the aligned 16-byte-per-lane stream, on a known byte count in this very pattern: the same kernels over 6e6 packed
        # lines of 0-1024 bytes with early retire OFF (every byte of every line is fetched): true bytes / FETCH_SIZE = 1.067 (column
        # table) and 1.135 (C3 table) on the final kernels (profiles/r06u_ragged_fetch_calib.txt; 1.135 / 1.319 on the 8- / 10-wave
        # kernels of r06n: the tally moves with the kernel's timing); whole aligned 1 KiB rows through the same kernel: 2.000.
        # With early retire ON (the default) a line that can no longer change state is not read on: traffic < algorithmic is real then.
        calib = {"c2": 1.067, "c3": 1.135}.get(w_)
        if calib is not None:
            read = raw * calib
    fetch_derived = read + write_kib * 1024
    if req is not None:
        # round 5: the read side from the request-size counters of the same launches -- derived without assuming that the kernel
        # fetches exactly its algorithmic bytes (the round-4 calibration factor did: it is kept for comparison only)
        read = req["read_bytes_by_request_size"]
    hbm = read + write_kib * 1024
    summary = {
        "command": f"rocprofv3 --kernel-trace --stats -- python bench.py --workload {wl} --steps 5 --warmup 2 --no-cpu-baseline (FSM_BENCH_LINES_FORMS=off64; then --pmc FETCH_SIZE, --pmc WRITE_SIZE and --pmc TCC_EA0_RDREQ_{{32B,64B,128B}} in separate passes)",
        "bench_line_under_trace": bench,
        "kernel_stats_top": [{"name": t["name"][:120], "calls": t["total_calls"], "avg_us": round(t["average"], 1), "pct": round(t["percentage"], 2)} for t in top[:5]],
        "walk_kernel": {"name": walk["name"], "calls": walk["total_calls"], "avg_ms": round(walk["average"] / 1e3, 4),
                        "algorithmic_GBps": round(alg / (walk["average"] * 1e-6) / 1e9, 1)},
        "pmc": {"FETCH_SIZE_KiB_per_launch": pmc["FETCH_SIZE"], "WRITE_SIZE_KiB_per_launch": pmc["WRITE_SIZE"], "fetch_bytes_raw": raw, "fetch_bytes_doubled": 2 * raw,
                "read_bytes_taken": read, "read_requests_by_size": req, "hbm_bytes_from_FETCH_SIZE_with_round4_factor": fetch_derived, "fetch_calibration_factor": calib, "bytes_that_must_be_read": must_read, "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
                "traffic_over_algorithmic": round(hbm / alg, 4)},
    }
    json.dump(summary, open(out + "_rocprof_summary.json", "w"), indent=1)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench as B
    json.dump({"n": n, "hbm_bytes_per_launch": hbm, "kernel": walk["name"], "kernels_sha16": B.kernels_sha16(), "source": os.path.basename(out) + "_rocprof_summary.json"},
              open(os.path.join(os.path.dirname(out), f"pmc_{w_}_{kind}.json"), "w"))
    print(json.dumps(summary["walk_kernel"]), json.dumps(summary["pmc"]["traffic_over_algorithmic"]))


def main():
    prof, wl, out = sys.argv[1:4]
    # the full record of the traced run (bench.py prints a compact line on stdout and writes the whole thing next to it)
    detail = os.path.join(prof, "bench_detail_trace.json")
    bench = json.load(open(detail)) if os.path.exists(detail) else json.loads(open(os.path.join(prof, "bench_trace.json")).read().strip().splitlines()[-1])
    top = rows(glob.glob(os.path.join(prof, "trace", "*.db"))[0], "select name, total_calls, total_duration, average, percentage from top_kernels")
    walk = sorted([t for t in top if "walk_" in t["name"]], key=lambda t: -t["total_duration"])[0]   # (the device-side choice launches two: the one that ran)
    pmc = {}
    for which, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        db = glob.glob(os.path.join(prof, which, "*.db"))[0]
        r = rows(db, f"select name, counter_name, counter_value, duration from pmc_events where counter_name = '{ctr}' and name = ?", (walk["name"],))
        if not r:
            r = rows(db, f"select name, counter_name, counter_value, duration from pmc_events where counter_name = '{ctr}' and name like '%walk_%'")
        pmc[ctr] = [x["counter_value"] for x in r]
    # the fabric read requests by size (profile.sh pass 4): bytes = 32 a + 64 b + 128 c -- no assumption about the access pattern
    req = None
    rdb = glob.glob(os.path.join(prof, "req", "*.db"))
    if rdb:
        r = rows(rdb[0], "select counter_name, counter_value from pmc_events where name = ?", (walk["name"],))
        if not r:
            r = rows(rdb[0], "select counter_name, counter_value from pmc_events where name like '%walk_%'")
        acc = {}
        for x in r:
            acc.setdefault(x["counter_name"].replace("_sum", ""), []).append(x["counter_value"])
        if acc:
            m = {k: sum(v) / len(v) for k, v in acc.items()}
            a32, a64, a128, tot = m.get("TCC_EA0_RDREQ_32B", 0.0), m.get("TCC_EA0_RDREQ_64B", 0.0), m.get("TCC_EA0_RDREQ_128B", 0.0), m.get("TCC_EA0_RDREQ", 0.0)
            req = {"RDREQ_32B": a32, "RDREQ_64B": a64, "RDREQ_128B": a128, "RDREQ": tot,
                   "read_bytes_by_request_size": 32.0 * a32 + 64.0 * a64 + 128.0 * a128,
                   "note": "TCC_EA0_RDREQ_{32B,64B,128B}: the L2's fabric read requests by size, per walk launch; bytes = 32 a + 64 b + 128 c "
                           "(if the three do not add up to RDREQ, the remainder is counted as 64-byte requests)"}
            rest = tot - (a32 + a64 + a128)
            if rest > 0.005 * max(tot, 1.0):
                req["read_bytes_by_request_size"] += 64.0 * rest
                req["unclassified_requests"] = rest
    n, L = bench["config"]["inputs_per_gpu"], bench["config"]["input_len"]
    fetch_kib = sum(pmc["FETCH_SIZE"]) / len(pmc["FETCH_SIZE"])
    write_kib = sum(pmc["WRITE_SIZE"]) / len(pmc["WRITE_SIZE"])
    if "lines" in bench["config"] and "_" in wl and not wl.endswith("eager40"):
        return lines_summary(prof, wl, out, bench, top, walk, pmc, fetch_kib, write_kib, req)
    # Read side.  gfx950 tallies a 128-byte request of a 16-byte-per-lane coalesced stream as 64 bytes
    # (MI355X_MICROARCH.md section HBM): the streamed input, n * L bytes, shows up as n * L / 2.  A gather that
    # misses is ONE 64-byte request tallied as 64 bytes (tools/fetch_calib.py, profiles/r02e_fetch_calib.json:
    # 63.99 bytes and 1.000 TCC_EA0_RDREQ per 4- or 16-byte gather), so whatever FETCH_SIZE reports beyond the
    # input's n * L / 2 is table traffic at face value.
    raw = fetch_kib * 1024
    stream_raw = min(raw, n * L / 2)
    hbm = stream_raw * 2 + (raw - stream_raw) + write_kib * 1024
    alg = bench["roofline"].get("algorithmic_bytes_per_launch") or n * (L + 4)
    fetch_derived = hbm
    if req is not None:
        hbm = req["read_bytes_by_request_size"] + write_kib * 1024
    summary = {
        "command": f"rocprofv3 --kernel-trace --stats -- python bench.py --workload {wl} --steps 5 --warmup 2 --no-cpu-baseline --subs none  (then --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes)",
        "bench_line_under_trace": bench,
        "kernel_stats_top": [{"name": t["name"][:120], "calls": t["total_calls"], "avg_us": round(t["average"], 1), "pct": round(t["percentage"], 2)} for t in top[:4]],
        "walk_kernel": {"name": walk["name"], "calls": walk["total_calls"], "avg_ms": round(walk["average"] / 1e3, 4),
                        "algorithmic_GBps": round(alg / (walk["average"] * 1e-6) / 1e9, 1)},
        "pmc": {"FETCH_SIZE_KiB_per_launch": pmc["FETCH_SIZE"], "WRITE_SIZE_KiB_per_launch": pmc["WRITE_SIZE"],
                "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round(hbm / alg, 4),
                "read_requests_by_size": req, "hbm_bytes_from_FETCH_SIZE_x2_rule": fetch_derived,
                "note": "read side = 2 x the input stream's share of FETCH_SIZE (gfx950 tallies 128-B requests of 16-B/lane streams as 64 B) + the rest of FETCH_SIZE at face value (gather misses are 64-B requests tallied as 64 B: profiles/r02e_fetch_calib.json); write side as reported"},
    }
    l2db = glob.glob(os.path.join(prof, "l2", "*.db"))
    if l2db:  # optional pass: L2 (TCC) request / hit / miss counts per walk launch
        r = rows(l2db[0], "select counter_name, counter_value from pmc_events where name like '%walk_%'")
        l2 = {}
        for x in r:
            l2.setdefault(x["counter_name"], []).append(x["counter_value"])
        summary["l2_per_launch"] = {k: sum(v) / len(v) for k, v in l2.items()}
        summary["l2_per_launch"]["input_bytes_per_launch"] = n * L
    json.dump(summary, open(out + "_rocprof_summary.json", "w"), indent=1)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench   # kernels_sha16(): the counters speak for the device sources they were taken from
    json.dump({"n": n, "len": L, "hbm_bytes_per_launch": hbm, "kernel": walk["name"], "kernels_sha16": bench.kernels_sha16(),
               "source": os.path.basename(out) + "_rocprof_summary.json"},
              open(os.path.join(os.path.dirname(out), f"pmc_{wl}.json"), "w"))
    print(json.dumps(summary["walk_kernel"]), json.dumps(summary["pmc"]["traffic_over_algorithmic"]))


if __name__ == "__main__":
    main()
