#!/usr/bin/env python3
"""tests/tools/lds2_probe.py -- the stride-2 LDS layout (one lookup per two input bytes) against the one-lookup-per-byte
layouts on mid-size dense DFAs with few byte classes: literal sets over a small alphabet (unanchored, end-ids: no
absorbing state), uniform random text over that alphabet, every 4th row ending in a word.  Every layout's end states must
equal the first one's, and the first 512 rows the oracle's.  (Under tests/: the oracle is the checker.)"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8_000_000)
    ap.add_argument("--len", type=int, default=1024)
    ap.add_argument("--cases", default="abcdefghijkl:70,abcdefg:200,abcdefghijklmnopqrst:25")
    ap.add_argument("--layouts", default="9,2,5,8,6,0")
    a = ap.parse_args()
    import torch
    import libfsm_amd as hip
    from oracle.pyoracle import Oracle
    hip.load_library()
    torch.cuda.set_device(0)
    n, L = a.n, a.len
    buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
    end = torch.empty(n, dtype=torch.int32, device="cuda")
    for case in a.cases.split(","):
        alpha_s, nw = case.split(":")
        alpha_b = alpha_s.encode()
        rng = np.random.RandomState(len(alpha_b) * 131 + int(nw))
        al = np.frombuffer(alpha_b, np.uint8)
        words = sorted(set(bytes(al[rng.randint(0, len(al), rng.randint(3, 8))]) for _ in range(int(nw))))
        flat = hip.FlatDfa.from_strings(words, 0, list(range(len(words))))
        hip.gen_inputs_device(buf.data_ptr(), n, L, 0, 11, alpha_b, None, 0)
        torch.cuda.synchronize()
        want = Oracle(flat).table_walk(buf[:512].cpu().numpy())
        ref = None
        for layout in [int(x) for x in a.layouts.split(",")]:
            try:
                dfa = hip.HipDfa(flat, layout)
            except OSError:
                print(f"{alpha_s:22s} states={flat.nstates:5d} layout {layout}: does not hold this DFA", flush=True)
                continue
            info = dfa.info()
            ms = []
            for _ in range(5):
                dfa.exec_batch_device(buf.data_ptr(), L, n, end.data_ptr(), 0)
                ms.append(dfa.last_kernel_ms())
            torch.cuda.synchronize()
            ok = np.array_equal(end[:512].cpu().numpy().view(np.uint32), want)
            if ref is None:
                ref = end.clone()
            ok = ok and bool(torch.equal(ref, end))
            print(f"{alpha_s:22s} states={flat.nstates:5d} classes={info['nclasses']:3d} layout={'auto:' if layout == 0 else ''}{info['layout_name']:8s} table={info['table_bytes']:7d} B "
                  f"waves={info['waves_per_block']:2d} {min(ms[1:]):8.3f} ms {n * L / min(ms[1:]) / 1e6:8.1f} GB/s  {dfa.last_kernel_name()}  {'ok' if ok else 'MISMATCH'}", flush=True)
            dfa.close()


if __name__ == "__main__":
    main()
