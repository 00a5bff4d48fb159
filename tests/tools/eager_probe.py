#!/usr/bin/env python3
"""tests/tools/eager_probe.py -- throughput of the eager-output walk on the kind of DFA
it exists for: fsm_union_repeated_pattern_group over K unanchored literal patterns (one eager id each),
built by the real reference; random lowercase text with a pattern planted in every 4th input.
(Under tests/: uses the reference to build the automaton and the oracle as checker.)"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", default="40,150")
    ap.add_argument("--n", type=int, default=2_000_000)
    ap.add_argument("--len", type=int, default=1024)
    ap.add_argument("--layouts", default="0", help="comma list: 0 auto, 2 lds, 8 ldsself, 4 global")
    ap.add_argument("--waves", default="0", help="comma list of KNOB_WAVES values (0 = the library's choice)")
    ap.add_argument("--modes", default="-1", help="comma list of KNOB_INPUT_MODE values (-1 = the library's choice, 1 = LDS-DMA, 0 = per-lane loads)")
    a = ap.parse_args()
    import torch
    import libfsm_amd as hip
    from oracle.pyoracle import Oracle, RefFsm
    hip.load_library()
    torch.cuda.set_device(0)
    n, L = a.n, a.len
    alpha = b"abcdefghijklmnopqrstuvwxyz"
    buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
    end = torch.empty(n, dtype=torch.int32, device="cuda")
    for K in [int(x) for x in a.k.split(",")]:
        rng = np.random.RandomState(K)
        al = np.frombuffer(alpha, np.uint8)
        words = sorted(set(bytes(al[rng.randint(0, 26, rng.randint(4, 8))]) for _ in range(2 * K)))[:K]
        f = RefFsm.union_repeated("pcre", words, 1, False)
        flat = f.flatten()
        hip.gen_inputs_device(buf.data_ptr(), n, L, 0, 7, alpha, words[0], 4)
        torch.cuda.synchronize()
        k = 256
        _, wend, wsets = Oracle(flat).exec_eager(buf[:k].cpu().numpy(), None, cap=K + 8)
        for layout in [int(x) for x in a.layouts.split(",")]:
            dfa = hip.HipDfa(flat, layout)
            info = dfa.info()
            W = dfa.eager_words()
            sets = torch.zeros((n, W), dtype=torch.int64, device="cuda")
            for md, wv, (name, fn) in [(int(m), int(w), nf) for m in a.modes.split(",") for w in a.waves.split(",") for nf in (
                    ("plain walk", lambda: dfa.exec_batch_device(buf.data_ptr(), L, n, end.data_ptr(), 0)),
                    ("eager walk", lambda: dfa.exec_batch_eager_device(buf.data_ptr(), L, n, end.data_ptr(), sets.data_ptr())))]:
                dfa.tune(hip.KNOB_WAVES, wv)
                dfa.tune(hip.KNOB_INPUT_MODE, md)
                name = f"{name} mode={md} waves={wv}"
                ms = []
                for _ in range(4):
                    fn()
                    ms.append(dfa.last_kernel_ms())
                torch.cuda.synchronize()
                t = min(ms[1:])
                print(f"K={K:4d} states={flat.nstates:6d} layout={info['layout_name']:8s} words/input={W} {name}: {t:8.3f} ms  "
                      f"{n * L / t / 1e6:8.1f} GB/s", flush=True)
            ids = np.array([dfa._lib.fsm_hip_eager_id(C.c_void_p(dfa._h), b) for b in range(dfa.eager_id_count())], np.uint32)
            bits = np.unpackbits(sets[:k].cpu().numpy().view(np.uint8).reshape(k, W * 8), axis=1, bitorder="little")[:, :len(ids)].astype(bool)
            ok = np.array_equal(end[:k].cpu().numpy().view(np.uint32), wend) and all(np.array_equal(ids[bits[i]], wsets[i]) for i in range(k))
            print(f"K={K:4d} layout={info['layout_name']:8s} first {k} inputs vs oracle: {'OK' if ok else 'MISMATCH'}; "
                  f"inputs with outputs: {int((sets != 0).any(dim=1).sum())}", flush=True)
            dfa.close()


if __name__ == "__main__":
    main()
