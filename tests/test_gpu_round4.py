"""GPU (-m gpu), round 4: the lazy walk of the sparse layout (walk_lazy.h), the compact-metadata batch fronts
(u32 offsets, lengths only), the stride-2 LDS layout.  Everything through the C ABI, compared with the oracle (the
plain-C restatement of fsm_exec) or the committed golden answers of the real reference -- never with itself."""
import os

import numpy as np
import pytest

from common import GOLDEN, Golden

pytestmark = pytest.mark.gpu

NO = 0xFFFFFFFF
KNOB_SPARSE_FAST = 20


@pytest.fixture(scope="module")
def hip(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    torch.cuda.set_device(0)
    import libfsm_amd
    libfsm_amd.load_library()   # raises if the HIP extension is missing: no silent fallback
    return libfsm_amd


def literal_set(hip, rng, alpha_b, nwords, lo, hi, flags, endids=True):
    alpha = np.frombuffer(alpha_b, np.uint8)
    words = sorted(set(bytes(alpha[rng.randint(0, len(alpha), rng.randint(lo, hi + 1))]) for _ in range(nwords)))
    return words, hip.FlatDfa.from_strings(words, flags, list(range(len(words))) if endids else None)


def rows_over(rng, alpha_b, n, L, words, every=3, foreign=0.0):
    alpha = np.frombuffer(alpha_b, np.uint8)
    rows = alpha[rng.randint(0, len(alpha), (n, L))]
    if foreign > 0:                                  # bytes outside the alphabet: classes that own no bit
        m = rng.rand(n, L) < foreign
        rows[m] = rng.randint(0, 256, int(m.sum())).astype(np.uint8)
    for i in range(0, n, every):
        w = words[rng.randint(len(words))]
        if len(w) <= L:
            rows[i, L - len(w):] = np.frombuffer(w, np.uint8)
    return rows


@pytest.mark.parametrize("shape", [
    (b"abcd", 400, 3, 9, 2, 64 * 1024),                                  # three full levels in LDS, deep chains below them
    (b"abcdefgh", 3000, 4, 10, 2, 64 * 1024),
    (b"abcdefghijklmnop", 4000, 5, 9, 0, 0),                             # unanchored, end-ids: accept states are not absorbing
    (b"abcdefghijklmnopqrstuvwxyz0123456789", 3000, 6, 12, 2, 0),        # depth-1 not full: most answers are sentinels
    (b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_-", 20000, 8, 16, 2, 0),   # configs[4]'s alphabet: 64 bits + a class without one
], ids=["a4", "a8", "a16-unanchored", "a36", "a64"])
def test_lazy_walk_literal_sets(hip, shape):
    """walk_lazy on literal sets of several shapes, forced (KNOB_SPARSE_FAST = 3) whatever the default would be: rows over
    the alphabet, rows with foreign bytes (classes without a bit: the exact path), batch sizes around the 128-input tile,
    64-byte to 1 KiB rows.  Against the oracle's table walk."""
    from oracle.pyoracle import Oracle
    from libfsm_amd import Plan
    alpha_b, nw, lo, hi, flags, _lds = shape
    rng = np.random.RandomState(len(alpha_b) * 7 + nw)
    words, flat = literal_set(hip, rng, alpha_b, nw, lo, hi, flags)
    has_lazy = Plan(flat, hip.LAYOUT_SPARSE).get("lazy").size > 0
    assert has_lazy, "the planner made no lazy form for this literal set"
    orc = Oracle(flat)
    dfa = hip.HipDfa(flat, hip.LAYOUT_SPARSE)
    for n, L, foreign in ((1, 64, 0.0), (127, 64, 0.0), (129, 128, 0.0), (1000, 192, 0.0), (3001, 1024, 0.0), (2500, 256, 0.02), (640, 64, 0.3)):
        rows = rows_over(rng, alpha_b, n, L, words, foreign=foreign)
        want = orc.table_walk(rows)
        for knob in (3, 1, 0):                       # lazy, record-as-state, chain loop: all three must agree with the oracle
            dfa.tune(KNOB_SPARSE_FAST, knob)
            end, bm = dfa.exec_batch(rows, want_bitmap=True)
            assert np.array_equal(end, want), (n, L, foreign, knob, int((end != want).sum()))
            assert np.array_equal(np.unpackbits(bm.view(np.uint8), bitorder="little")[:n].astype(bool), want != NO)
    dfa.close()


def test_lazy_walk_kernel_variants(hip):
    """The tile counter and static striding, on odd batch sizes (the last tile partial, fewer tiles than wavefronts, one
    input; 32- and 96-byte rows go to the record-as-state walk: the lazy one takes multiples of 64)."""
    from oracle.pyoracle import Oracle
    rng = np.random.RandomState(77)
    alpha_b = b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_-"
    words, flat = literal_set(hip, rng, alpha_b, 20000, 8, 16, 2)
    orc = Oracle(flat)
    dfa = hip.HipDfa(flat, hip.LAYOUT_SPARSE)
    dfa.tune(KNOB_SPARSE_FAST, 3)
    for n, L in ((1, 32), (191, 96), (193, 1024), (70001, 160)):
        rows = rows_over(rng, alpha_b, n, L, words, foreign=0.001)
        want = orc.table_walk(rows)
        for dyn in (1, 0):
            dfa.tune(21, dyn)
            for rep in range(3):
                end, bm = dfa.exec_batch(rows)
                assert np.array_equal(end, want), (n, L, dyn, rep, int((end != want).sum()))
                assert np.array_equal(np.unpackbits(bm.view(np.uint8), bitorder="little")[:n].astype(bool), want != NO)
    dfa.close()


def test_lazy_walk_absorbing_accept(hip):
    """An unanchored literal set WITHOUT end-ids: every output node collapses into one absorbing accept state (re_strings
    semantics, ac.c:293-296): the kernel variant that tests for absorbing states, with and without the wave retire."""
    from oracle.pyoracle import Oracle
    rng = np.random.RandomState(11)
    alpha_b = b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_-"
    words, flat = literal_set(hip, rng, alpha_b, 30000, 6, 12, 0, endids=False)
    orc = Oracle(flat)
    dfa = hip.HipDfa(flat, hip.LAYOUT_SPARSE)
    rows = rows_over(rng, alpha_b, 5000, 512, words, every=2)
    want = orc.table_walk(rows)
    assert 1000 < (want != NO).sum() < 5000
    for early in (1, 0):
        dfa.tune(hip.KNOB_EARLY_RETIRE, early)
        for knob in (3, 1):
            dfa.tune(KNOB_SPARSE_FAST, knob)
            end, _ = dfa.exec_batch(rows)
            assert np.array_equal(end, want), (early, knob)
    dfa.close()


def test_lazy_walk_is_the_default_for_configs4(hip):
    """BASELINE configs[4] at a tenth of its size (1e4 literals of 8-16 symbols over 64): AUTO picks the sparse layout and the
    lazy walk is what runs; 2e4 rows of 1 KiB against the oracle, and repeated launches agree with each other."""
    from oracle.pyoracle import Oracle
    from libfsm_amd import Plan
    rng = np.random.RandomState(1234)
    alpha_b = b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_-"
    words, flat = literal_set(hip, rng, alpha_b, 40000, 8, 16, 2)
    pl = Plan(flat)
    assert pl.layout == hip.LAYOUT_SPARSE
    z = pl.get("lazy")
    assert z.size > 0 and (int(z[9]) + int(z[10])) * 10 <= pl.abs_min - int(z[1]), "the lazy walk should be the default here"
    dfa = hip.HipDfa(flat)
    rows = rows_over(rng, alpha_b, 20000, 1024, words, every=8)
    want = Oracle(flat).table_walk(rows)
    assert (want != NO).sum() >= 2000
    first = None
    for rep in range(5):
        end, _ = dfa.exec_batch(rows)
        assert np.array_equal(end, want), rep
        first = end if first is None else first
        assert np.array_equal(end, first)
    dfa.tune(KNOB_SPARSE_FAST, 1)
    end, _ = dfa.exec_batch(rows)
    assert np.array_equal(end, want)
    dfa.close()
