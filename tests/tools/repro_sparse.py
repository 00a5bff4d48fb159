#!/usr/bin/env python3
"""tests/tools/repro_sparse.py -- the harness that showed one build of walk_lines32<SparsePol> losing a lane's state between an
input's first and second chunk (profiles/r08i_lines32_sparse_intermittent.txt): the `len1to40` case of
tests/test_gpu_round3.py::test_packed_kernel_length_distributions on the C1 automaton with the sparse layout forced, REPS
launches per row with and without the accept bitmap, by wavefronts per workgroup.  The record walk is kept off that kernel
(launch.h lines32_ok), so on the shipped tree every row reads 0 -- the harness stays for whoever takes the question up again
(flip lines32_ok<SparsePol> and rebuild)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import libfsm_amd as hip
    hip.load_library()
    from common import GOLDEN, Golden
    import test_gpu_round3 as t3
    from oracle.pyoracle import Oracle
    name = "c1.npz"
    reps = int(os.environ.get("REPS", 30))
    rng = np.random.RandomState(5 + len(name))
    g = Golden(os.path.join(GOLDEN, name))
    strings = t3._cases(name, rng)["len1to40"]
    ret, want = Oracle(g.flat).exec_strings(strings)
    base, off = t3._packed(strings)
    dfa = hip.HipDfa(g.flat, hip.LAYOUT_SPARSE)
    dfa.tune(hip.KNOB_INPUT_MODE, 2)
    total_bad = 0
    for waves in (0, 16, 12, 8, 4, 1):
        dfa.tune(hip.KNOB_WAVES, waves)
        bad_a = bad_b = 0
        for _ in range(reps):
            end, _bm = dfa.exec_batch_offsets(base, off)
            bad_a += int(not np.array_equal(end, want))
            end, _bm = dfa.exec_batch_offsets(base, off, want_bitmap=False)
            b = np.nonzero(end != want)[0]
            bad_b += int(len(b) != 0)
            if len(b) and bad_b <= 2:
                print("   wrong lines", b[:6], "lengths", [len(strings[i]) for i in b[:6]])
        total_bad += bad_a + bad_b
        print(f"waves {waves:2d}: wrong launches with bitmap {bad_a}/{reps}, end states only {bad_b}/{reps}  [{dfa.last_kernel_name()[:56]}]", flush=True)
    dfa.close()
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()
