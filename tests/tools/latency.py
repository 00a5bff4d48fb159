#!/usr/bin/env python3
"""tests/tools/latency.py -- cost of ONE host-pointer call (the retest / re(1) usage: a launch per input
line): fsm_hip_match_buffer on one short string, fsm_hip_exec_batch on n = 1, 64, 4096 rows of 256 B."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import ctypes as C
    import libfsm_amd as hip
    lib = hip.load_library()
    flat = hip.FlatDfa.load(os.path.join(ROOT, "tests", "golden", "c1.npz"))
    dfa = hip.HipDfa(flat)
    s = b"xxLibfsmsmyy"
    lib.fsm_hip_match_buffer.restype = C.c_int
    for _ in range(20):
        assert lib.fsm_hip_match_buffer(C.c_void_p(dfa._h), s, len(s)) == 1
    k = 2000
    t = time.perf_counter()
    for _ in range(k):
        lib.fsm_hip_match_buffer(C.c_void_p(dfa._h), s, len(s))
    dt = time.perf_counter() - t
    print(f"fsm_hip_match_buffer(12 B): {dt / k * 1e6:8.1f} us/call")
    for n in (1, 64, 4096, 65536):
        rows = hip.gen_inputs_host(n, 256, 0, 1, None, b"Libfsm", 8)
        for _ in range(5):
            end, _ = dfa.exec_batch(rows)
        k = 300
        t = time.perf_counter()
        for _ in range(k):
            dfa.exec_batch(rows)
        dt = time.perf_counter() - t
        print(f"fsm_hip_exec_batch(n={n:6d} x 256 B): {dt / k * 1e6:8.1f} us/call  {n * 256 * k / dt / 1e9:7.3f} GB/s")


if __name__ == "__main__":
    main()
