"""scratch: reproduce test_packed_kernel_length_distributions[c1] layout 7 mode -1 end-only"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import libfsm_amd as hip
hip.load_library()
from common import GOLDEN, Golden
import test_gpu_round3 as t3
from oracle.pyoracle import Oracle
name = "c1.npz"
rng = np.random.RandomState(5 + len(name))
g = Golden(os.path.join(GOLDEN, name))
o = Oracle(g.flat)
cases = t3._cases(name, rng)
strings = cases["len1to40"]
ret, want = o.exec_strings(strings)
base, off = t3._packed(strings)
for L in (7,):
    try:
        dfa = hip.HipDfa(g.flat, L)
    except OSError:
        continue
    for early in (1, 1 | 2048, 1, 1 | 2048):
        dfa.tune(hip.KNOB_EARLY_RETIRE, early)
        for mode, wv, bpc in ((2, 0, 0), (2, 12, 1)):
            dfa.tune(hip.KNOB_INPUT_MODE, mode)
            dfa.tune(hip.KNOB_WAVES, wv)
            dfa.tune(hip.KNOB_BLOCKS_PER_CU, bpc)
            bad_a = bad_b = 0
            for rep in range(30):
                end, bm = dfa.exec_batch_offsets(base, off)
                ka = dfa.last_kernel_name()
                bad_a += int(not np.array_equal(end, want))
                end, bm = dfa.exec_batch_offsets(base, off, want_bitmap=False)
                kb = dfa.last_kernel_name()
                b = np.nonzero(end != want)[0]
                bad_b += int(len(b) != 0)
                if len(b) and bad_b <= 2:
                    print("   bad idx", b[:6], "n", len(strings), "lens", [len(strings[i]) for i in b[:6]], "got", end[b[:6]], "want", want[b[:6]])
            print(f"layout {L} early {early} mode {mode} waves {wv} bpc {bpc}: with bitmap bad {bad_a}/30 [{ka[:50]}], end only bad {bad_b}/30 [{kb[:50]}]", flush=True)
    dfa.close()
