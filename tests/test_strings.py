"""fsm_hip_strings_*: the literal-set builder must produce the automaton libre's re_strings builds
(src/libre/re_strings.c, src/libre/ac.c), state for state: same ids, edges, end flags, end-id sets."""
import glob
import os
import random

import numpy as np
import pytest

from common import GOLDEN, Golden, ac_golden_paths, golden_id
from libfsm_amd import FlatDfa
from oracle.pyoracle import RefFsm, have_ref


def same(a: FlatDfa, b: FlatDfa):
    ca, cb = a.canonical(), b.canonical()
    return ca[0] == cb[0] and ca[1] == cb[1] and all(np.array_equal(x, y) for x, y in zip(ca[2:], cb[2:]))


@pytest.mark.parametrize("k", [1, 2, 3])
def test_golden_re_strings_constructions(k):
    # tests/golden/re_strings_<k>.npz: flattened from the reference's own re_strings (flags 0, end-id = index)
    g = Golden(os.path.join(GOLDEN, f"re_strings_{k}.npz"))
    words = [w.encode() for w in g.meta["words"]]
    assert same(FlatDfa.from_strings(words, g.meta["flags"], list(range(len(words)))), g.flat)


def test_recorded_re_strings_programs():
    # automata logged from the reference's own tests/re_strings/*.c programs (testutil.c:15-36: flags 0,
    # end-id = position in the list).  Every word of a program is an accepted input of its log, with the
    # complete end-id set of its node, so the (word, id) pairs rebuild the program's word list.
    paths = sorted(glob.glob(os.path.join(GOLDEN, "recorded", "re_strings_*.npz")))
    assert len(paths) >= 3
    for p in paths:
        g = Golden(p)
        assert (g.ret == 1).all()
        words, ids = [], []
        for i, w in enumerate(g.strings()):
            for e in g.ids_of(i):
                words.append(w)
                ids.append(int(e))
        assert same(FlatDfa.from_strings(words, 0, ids), g.flat), p


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_against_live_reference_small_sets():
    random.seed(20260924)
    n = 0
    for trial in range(120):
        nw = random.choice([0, 1, 2, 3, 5, 8, 20, 60])
        alpha = random.choice([b"ab", b"abc", b"abcdefgh", bytes(range(0, 256))])
        words = [bytes(random.choice(alpha) for _ in range(random.choice([0, 1, 1, 2, 3, 4, 6]))) for _ in range(nw)]
        for flags in range(8):
            for with_ids in (True, False):
                ref = RefFsm.re_strings(words, flags, with_ids).flatten()
                mine = FlatDfa.from_strings(words, flags, list(range(len(words))) if with_ids else None)
                assert same(ref, mine), (words, flags, with_ids)
                n += 1
    assert n == 120 * 16


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_against_live_reference_suffix_overlaps_and_shared_ids():
    # words that are suffixes / infixes of each other exercise the output propagation of ac.c:243
    sets = [
        [b"abc", b"b", b"bc", b"c"],
        [b"he", b"she", b"his", b"hers"],
        [b"aaaa", b"aa", b"a"],
        [b"xabcx", b"abc", b"bcx", b"cx", b"x"],
        [b"ab", b"bab", b"abab", b"b"],
    ]
    for words in sets:
        for flags in range(8):
            for ids in (None, list(range(len(words)))):   # the reference helper numbers end-ids by position
                g = RefFsm.re_strings(words, flags, ids is not None)
                assert same(g.flatten(), FlatDfa.from_strings(words, flags, ids)), (words, flags, ids)


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_against_live_reference_20k_words():
    rng = np.random.RandomState(11)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", np.uint8)
    words = [bytes(alpha[rng.randint(0, 26, rng.randint(3, 9))]) for _ in range(20000)]
    ref = RefFsm.re_strings(words, 2, True).flatten()
    mine = FlatDfa.from_strings(words, 2, list(range(len(words))))
    assert same(ref, mine)


def test_errors_and_reuse():
    import ctypes as C
    from libfsm_amd.capi import load_library
    lib = load_library()
    g = lib.fsm_hip_strings_new()
    assert g
    assert lib.fsm_hip_strings_add_raw(g, b"abc", 3, None) == 1
    C.set_errno(0)
    assert not lib.fsm_hip_strings_build(g, 8)             # unknown flag
    d1 = lib.fsm_hip_strings_build(g, 0)
    assert d1 and d1.contents.nstates == 4                 # end + "", "a", "ab" ("abc" collapses into the end state)
    assert lib.fsm_hip_strings_add_raw(g, b"abd", 3, None) == 1
    d2 = lib.fsm_hip_strings_build(g, 2)                   # reusable; anchored right keeps every node
    assert d2.contents.nstates == 5
    lib.fsm_hip_desc_free(d1)
    lib.fsm_hip_desc_free(d2)
    lib.fsm_hip_strings_free(g)
    lib.fsm_hip_strings_free(None)


@pytest.mark.parametrize("path", ac_golden_paths(), ids=golden_id)
def test_aho_corasick_language_equals_the_regex(path):
    """The reference's tests/aho_corasick claim, on the frozen regex side: the literal set of in<n>.txt, built
    with the matching anchor flags, accepts exactly the strings the regex accepts -- every string up to the
    golden's length bound (oracle walk of the builder's DFA vs the reference's fsm_exec on the regex DFA)."""
    from oracle.pyoracle import Oracle
    g = Golden(path)
    words = [w.encode() for w in g.meta["words"]]
    flat = FlatDfa.from_strings(words, g.meta["strings_flags"], None)
    base, off = g.packed()
    ret, _ = Oracle(flat).exec_offsets(base, off)
    assert np.array_equal(ret == 1, g.ret == 1), g.meta
    assert 0 < int((g.ret == 1).sum()) < len(g.ret)
