#!/usr/bin/env python3
"""tests/tools/multi_timing.py -- the many-DFA front against what it replaces, on the reference's own retest corpus
(37 automata / 115 lines, tests/golden/retest; src/retest/main.c:1056-1058 + :1114 is the loop being replaced).

  A. library level (one process, warm HIP context), per pass over the 37 records, table build included:
       before: for each record  fsm_hip_dfa_create (upload)  + fsm_hip_exec_batch_offsets (one launch)  + free
       after : for each record  fsm_hip_dfa_create(FSM_HIP_DEFER_UPLOAD) ; ONE fsm_hip_exec_multi ; free
  B. the patched retest binary (integration/_build/retest), wall time of the whole process (fork, HIP context, regex
     compile on the CPU included) for -l hip (file per submission), -l hip-record (record per launch), -l hip-line, -l vm.
Prints one JSON object."""
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    import torch
    torch.cuda.set_device(0)
    import libfsm_amd as hip
    from common import Golden, all_golden_paths, retest_tst_lines
    hip.load_library()
    gs = [Golden(p) for p in all_golden_paths() if "/retest/" in p]
    jobs = [g.strings() for g in gs]
    want = [np.where(g.ret == 1, g.end, 0xFFFFFFFF).astype(np.uint32) for g in gs]
    out = {"records": len(gs), "lines": sum(len(j) for j in jobs)}

    def before():
        for g, j in zip(gs, jobs):
            d = hip.HipDfa(g.flat)
            end, _ = d.exec_strings(j)
            d.close()
        return end

    def after():
        ds = [hip.HipDfa(g.flat, hip.DEFER_UPLOAD) for g in gs]
        outs = hip.exec_multi(ds, jobs)
        for d in ds:
            d.close()
        return outs

    outs = after()
    assert all(np.array_equal(e, w) for (e, _), w in zip(outs, want))
    out["after_launches"] = hip.multi_last_launches()
    for name, fn in (("before_ms", before), ("after_ms", after)):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        out[name] = {"min": round(min(ts), 3), "median": round(statistics.median(ts), 3)}
    # the submission alone (automata already planned): ONE copy in, one kernel, one copy out
    ds = [hip.HipDfa(g.flat, hip.DEFER_UPLOAD) for g in gs]
    ds_up = [hip.HipDfa(g.flat) for g in gs]
    ts, tb = [], []
    for _ in range(3):
        hip.exec_multi(ds, jobs)
    for _ in range(50):
        t0 = time.perf_counter()
        hip.exec_multi(ds, jobs)
        ts.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        for d, j in zip(ds_up, jobs):
            d.exec_strings(j)
        tb.append((time.perf_counter() - t0) * 1e3)
    out["submit_only_ms"] = {"multi_min": round(min(ts), 3), "multi_median": round(statistics.median(ts), 3),
                             "one_by_one_min": round(min(tb), 3), "one_by_one_median": round(statistics.median(tb), 3)}
    out["speedup_with_table_build"] = round(out["before_ms"]["median"] / out["after_ms"]["median"], 2)
    out["speedup_submit_only"] = round(statistics.median(tb) / statistics.median(ts), 2)

    exe = os.path.join(ROOT, "integration", "_build", "retest")
    if os.path.exists(exe):
        lines, _ = retest_tst_lines()
        tst = "/tmp/multi_timing_all.tst"
        open(tst, "wb").write(("\n".join(lines) + "\n").encode("latin1"))
        env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
        out["retest"] = {}
        for impl in ("hip", "hip-record", "hip-line", "vm"):
            ts, batch = [], None
            for _ in range(5):
                t0 = time.perf_counter()
                r = subprocess.run([exe, "-l", impl, tst], capture_output=True, text=True, errors="replace", env=env, timeout=300)
                ts.append((time.perf_counter() - t0) * 1e3)
                assert r.returncode == 0 and r.stdout.count("[OK    ]") == 115, (impl, r.stdout[-500:], r.stderr[-500:])
                b = [l for l in r.stdout.splitlines() if l.startswith("[BATCH ]")]
                batch = b[0] if impl == "hip" and b else batch
            out["retest"][impl] = {"process_wall_ms_min": round(min(ts), 1), "process_wall_ms_median": round(statistics.median(ts), 1)}
            if batch:
                out["retest"][impl]["batch_line"] = batch
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
