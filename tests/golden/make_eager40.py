#!/usr/bin/env python3
"""tests/golden/make_eager40.py -- freezes the automaton of bench.py's `eager40` sub-result: the real reference's
fsm_union_repeated_pattern_group (include/fsm/bool.h:55-75) over 40 unanchored literal patterns of 4-7 lowercase letters,
one eager-output id each (exec.c:126-144 fires them), flattened through the product's own shim.  Run in the build container
(needs oracle/_ref, i.e. /root/reference at build time); the GPU box only reads tests/golden/bench/eager40.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.pyoracle import RefFsm  # noqa: E402

K = 40
rng = np.random.RandomState(K)
al = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", np.uint8)
words = sorted(set(bytes(al[rng.randint(0, 26, rng.randint(4, 8))]) for _ in range(2 * K)))[:K]
f = RefFsm.union_repeated("pcre", words, 1, False)
flat = f.flatten()
flat.save(os.path.join(ROOT, "tests", "golden", "bench", "eager40.npz"), patterns=np.frombuffer(b"\n".join(words), np.uint8))
print("eager40.npz:", flat.nstates, "states,", len(words), "patterns, eager ids:", sorted(set(flat.eager_ids.tolist()))[:5], "...")
