#!/usr/bin/env python3
"""tools/pmc_kernel.py -- per-kernel hardware counters for any command (run on the GPU box).

One rocprofv3 --pmc pass per counter set (sets separated by ';' -- 8 SQ slots, 4 TCC slots per pass; never
combined with a trace domain), then the mean of every counter per kernel name, read from the rocpd
databases.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md).
usage: pmc_kernel.py --sets "A B C;D E" [--match walk_] [--out file.json] -- <command ...>
"""
import argparse
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile


def main():
    if "--" not in sys.argv:
        sys.exit(__doc__)
    k = sys.argv.index("--")
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", required=True)
    ap.add_argument("--match", default="walk_")
    ap.add_argument("--out")
    a = ap.parse_args(sys.argv[1:k])
    cmd = sys.argv[k + 1:]
    cmd = [os.path.abspath(c) if os.path.exists(c) else c for c in cmd]     # rocprofv3 is run from /tmp
    env = dict(os.environ, TMPDIR="/tmp")
    res = {}
    for si, cs in enumerate(x.strip() for x in a.sets.split(";") if x.strip()):
        d = tempfile.mkdtemp(prefix="pmc", dir="/tmp")
        r = subprocess.run(["rocprofv3", "--pmc", *cs.split(), "-d", d, "-o", "p", "--"] + cmd, cwd="/tmp", env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(f"# set {si} ({cs}): rc={r.returncode}\n{r.stderr[-800:]}", file=sys.stderr)
        for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            con = sqlite3.connect(db)
            for name, ctr, val, n in con.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name"):
                if a.match in name:
                    short = name.split("(")[0].replace("void fsmhip::", "")
                    res.setdefault(short, {})[ctr] = val
                    res[short]["launches"] = n
            con.close()
        shutil.rmtree(d, ignore_errors=True)
    for name, c in sorted(res.items()):
        print(name)
        for ctr, v in sorted(c.items()):
            print(f"    {ctr:28s} {v:16.1f}")
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for ctr in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                        "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_LDS"):
                if ctr in c:
                    print(f"    {ctr + ' / SQ_WAVE_CYCLES':44s} {c[ctr] / wc:8.3f}")
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"command": " ".join(sys.argv[k + 1:]), "sets": a.sets, "kernels": res}, f, indent=1)


if __name__ == "__main__":
    main()
