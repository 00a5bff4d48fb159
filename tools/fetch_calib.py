#!/usr/bin/env python3
"""tools/fetch_calib.py -- what does FETCH_SIZE tally for a GATHER?

MI355X_MICROARCH.md calibrates FETCH_SIZE for 16-byte-per-lane coalesced streams (128-byte requests are
counted as 64 bytes: read side x2).  The sparse / global layouts add 4- and 16-byte gathers whose 64 lanes
touch 64 different lines, so before an HBM-traffic figure is quoted for them this tool measures the counter on
exactly that access shape: N independent gathers over a 16 GiB buffer (far beyond L2 + MALL, so nearly every
one is an HBM fetch).  Run it under rocprofv3, once per counter set (never with trace domains):
    rocprofv3 --pmc FETCH_SIZE -d out/fetch -o f -- python tools/fetch_calib.py
    rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d out/req -o r -- python tools/fetch_calib.py
then `python tools/fetch_calib.py --read out/fetch out/req` prints bytes and requests per gather.
"""
import ctypes as C
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 1 << 27          # gathers per launch
BYTES = 16 << 30     # buffer


def run():
    import torch
    import libfsm_amd as hip
    lib = hip.load_library()
    lib.fsm_hip_gather_probe_ms.restype = C.c_double
    buf = torch.empty(BYTES, dtype=torch.uint8, device="cuda")
    buf.fill_(1)
    scratch = torch.zeros(4, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for vec in (16, 4, 16, 4):
        ms = lib.fsm_hip_gather_probe_ms(C.c_void_p(buf.data_ptr()), C.c_size_t(BYTES), C.c_size_t(N), C.c_int(vec),
                                         C.c_void_p(scratch.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        print(f"gather_probe vec={vec:2d} n={N} ms={ms:.3f} Ggathers/s={N / ms / 1e6:.1f}", flush=True)


def read(dirs):
    out = {"gathers_per_launch": N, "buffer_bytes": BYTES}
    for d in dirs:
        for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            con = sqlite3.connect(db)
            for name, ctr, val in con.execute("select name, counter_name, counter_value from pmc_events where name like '%gather_probe%'"):
                vec = "16B" if "<16>" in name or "Li16" in name else "4B"
                out.setdefault(f"{ctr}_{vec}", []).append(val)
    res = {}
    for k, v in out.items():
        if isinstance(v, list):
            per = sum(v) / len(v) / N
            res[k + "_per_gather"] = round(per * (1024 if k.startswith("FETCH_SIZE") else 1), 3)   # FETCH_SIZE is in KiB
    out["per_gather"] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--read":
        read(sys.argv[2:])
    else:
        run()
