#!/usr/bin/env python3
"""tests/tools/file_probe.py -- fsm_hip_match_file on one big file (the c3u automaton: right-anchored patterns, every byte matters)
beside the time one host-to-device copy of the same bytes takes, and the reference's fsm_vm_match_file (VM v2, one host core) on
a prefix of it.  (Under tests/: the oracle / reference are the checkers.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import libfsm_amd as hip
    from common import GOLDEN, Golden
    from oracle.pyoracle import Oracle
    hip.load_library()
    size = int(os.environ.get("FILE_BYTES", 1 << 30))
    g = Golden(os.path.join(GOLDEN, "c3u.npz"))
    pats = bytes(np.load(os.path.join(GOLDEN, "c3u.npz"))["patterns"]).split(b"\n")
    rng = np.random.RandomState(3)
    alnum = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", np.uint8)
    data = alnum[rng.randint(0, 36, size)]
    suf = pats[7][:pats[7].index(b"[")] + b"4"
    data[size - len(suf):] = np.frombuffer(suf, np.uint8)
    path = "/tmp/fsm_file_probe.bin"
    data.tofile(path)
    dfa = hip.HipDfa(g.flat)
    dfa.match_buffer(b"warm up")
    for rep in range(3):
        t0 = time.perf_counter()
        r = dfa.match_file(path)
        t = time.perf_counter() - t0
        w, p = dfa.match_last_passes()
        print(f"fsm_hip_match_file: {size} bytes -> {r} in {t * 1e3:8.1f} ms = {size / t / 1e9:6.2f} GB/s  ({w} windows, {p} passes)", flush=True)
    pin = torch.from_numpy(data).pin_memory()
    dev = torch.empty(size, dtype=torch.uint8, device="cuda")
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print(f"one pinned host-to-device copy of the same bytes: {t * 1e3:8.1f} ms = {size / t / 1e9:6.2f} GB/s", flush=True)
    t0 = time.perf_counter()
    with open(path, "rb") as f:
        while f.read(32 << 20):
            pass
    t = time.perf_counter() - t0
    print(f"reading the file (page cache) in 32 MiB pieces: {t * 1e3:8.1f} ms = {size / t / 1e9:6.2f} GB/s", flush=True)
    o = Oracle(g.flat)
    k = min(size, 64 << 20)
    t0 = time.perf_counter()
    want = o.table_walk(data[None, size - k:])[0]
    t = time.perf_counter() - t0
    print(f"oracle table walk of the last {k} bytes, one host core: {t * 1e3:8.1f} ms = {k / t / 1e9:6.3f} GB/s (accepts: {int(want != 0xFFFFFFFF)})", flush=True)
    os.unlink(path)


if __name__ == "__main__":
    main()
