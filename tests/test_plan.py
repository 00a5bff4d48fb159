"""CPU: the host-side table planner (libfsm_amd/csrc/plan.cpp) encodes exactly the
DFA's transition function in every device layout.  The layouts are decoded here
with numpy (mirroring the kernels' step functions) and compared, for EVERY
(state, byte) pair, with the flat description's dense table."""
import numpy as np
import pytest

from common import Golden, all_golden_paths, golden_id
from libfsm_amd import (ALL_LAYOUTS, LAYOUT_LDS2, LAYOUT_COMB, LAYOUT_COMB256, LAYOUT_COMBSELF, LAYOUT_GLOBAL, LAYOUT_LDS, LAYOUT_LDSSELF, LAYOUT_SPARSE, LAYOUT_TINY,
                        FlatDfa, Plan)

NO = 0xFFFFFFFF


def reference_table(flat):
    """[S+1][256] in ORIGINAL numbering, missing edge -> S (dead)."""
    d = flat.dense().astype(np.int64)
    S = flat.nstates
    d[d == NO] = S
    return np.vstack([d, np.full((1, 256), S, np.int64)])


def decode_sparse(p, S1, lds_limit=160 * 1024):
    """SparsePol::next for every (state, byte), vectorised over the 256 bytes; the LDS mirror (records and
    dense rows of the first states) is read exactly where the kernel reads it."""
    img = p.get("sparse").astype(np.int64)
    assert img[0] == 0x31525053
    H, HDE, ldo, lro, gro, gdo, exo, N, Cn = (int(x) for x in img[1:10])
    ndense, nexc, maxchain = int(img[10]), int(img[11]), int(img[12])
    assert N == p.abs_min and Cn == p.C
    assert lro + 16 * H <= gro <= lds_limit and gro % 16 == 0          # the LDS part fits one workgroup
    pm = img[16:16 + 128]
    pm = np.stack([pm & 0xffff, pm >> 16], axis=1).reshape(-1)          # u16[256]
    cls_b, bit_b = pm & 0xff, pm >> 8
    assert np.array_equal(cls_b, p.get("cls").astype(np.int64))
    lds = img[:gro // 4]
    got = np.empty((S1, 256), np.int64)
    for n in range(S1):
        if n >= N:
            got[n] = n
            continue
        res = np.full(256, -1, np.int64)
        st = np.full(256, n, np.int64)
        live = np.ones(256, bool)
        hops = 0
        while live.any():
            idx = np.nonzero(live)[0]
            s0 = st[idx]
            rec = np.where((s0 < H)[:, None], lds[(lro // 4 + np.minimum(s0, H - 1) * 4)[:, None] + np.arange(4)] if H else 0,
                           img[(gro // 4 + s0 * 4)[:, None] + np.arange(4)])
            if H:
                in_lds = s0 < H
                assert np.array_equal(lds[(lro // 4 + s0[in_lds] * 4)[:, None] + np.arange(4)],
                                      img[(gro // 4 + s0[in_lds] * 4)[:, None] + np.arange(4)])
            dense = (rec[:, 2] & 0x80000000) != 0
            o = rec[:, 3] + cls_b[idx]
            dval = np.where(o < HDE, lds[np.minimum(ldo // 4 + o, len(lds) - 1)], img[np.minimum(gdo // 4 + o, len(img) - 1)])
            bits = rec[:, 0] | (rec[:, 1] << 32)
            b = bit_b[idx]
            sel = (b < 64) & (((bits >> np.minimum(b, 63)) & 1) == 1)
            low = bits & ((1 << np.minimum(b, 63)) - 1)
            pop = np.array([bin(int(x)).count("1") for x in low], np.int64)
            consec = (rec[:, 2] & 0x40000000) != 0                      # targets = first + rank: no list stored
            eval_ = np.where(consec, rec[:, 3] + pop, img[np.minimum(exo // 4 + rec[:, 3] + pop, len(img) - 1)])
            # FULLBASE: a miss resolves as first(base) + bit when the (LDS-resident, all-bits-set) base says so
            bid = rec[:, 2] & 0x0FFFFFFF
            fullbase = ((rec[:, 2] & 0x20000000) != 0) & ~dense & ~sel & (b < 64)
            assert (bid[fullbase] < H).all()
            fval = lds[np.minimum(lro // 4 + bid * 4 + 3, len(lds) - 1)] + b
            fin_now = dense | sel | fullbase
            res[idx[fin_now]] = np.where(sel, eval_, np.where(dense, dval, fval))[fin_now]
            st[idx[~fin_now]] = bid[~fin_now]
            live[idx[fin_now]] = False
            hops += 1
            assert hops <= maxchain + 1
        got[n] = res
    return got


def check_sparse_fast(p, got):
    """SparseFastPol::step_fast (walk_kernels.h) for every record it may be used on: the own record, its base B's and
    cf = first(base(B)) (B's base owning every bit) -- B and cf read as the kernel reads them (from the LDS mirror,
    unguarded) -- must give SparsePol's answer for every class that owns a bit whenever the kernel's own test lets the lane
    through (own hit: CONSEC; own miss: FAST).  The
    absorbing states' records (no bits, FAST) exist so that a state is entered without a range test."""
    img = p.get("sparse").astype(np.int64)
    H, HDE, ldo, lro, gro, gdo, exo, N, Cn = (int(x) for x in img[1:10])
    pm = img[16:16 + 128]
    pm = np.stack([pm & 0xffff, pm >> 16], axis=1).reshape(-1)
    bit_b = pm >> 8
    bytes_with_bit = np.nonzero(bit_b < 64)[0]
    lds = img[:gro // 4]
    nfast = 0

    def rec_g(n):
        return img[gro // 4 + n * 4: gro // 4 + n * 4 + 4]

    def rec_l(n):      # an LDS read cannot fault: out of range reads garbage
        o = lro // 4 + n * 4
        return lds[o:o + 4] if o + 4 <= len(lds) else np.zeros(4, np.int64)


    def probe(r, b):
        bits = int(r[0]) | (int(r[1]) << 32)
        return (bits >> b) & 1 == 1, int(r[3]) + bin(bits & ((1 << b) - 1)).count("1")

    assert (gdo - gro) // 16 == p.S1, "one record per state"
    for n in range(N, p.S1):
        assert list(rec_g(n)) == [0, 0, 0x10000000, 0]
    for n in range(N):
        ra = rec_g(n)
        B = int(ra[2]) & 0x0FFFFFFF
        rb = rec_l(B)
        cf = int(rec_l(int(rb[2]) & 0x0FFFFFFF)[3])
        for by in bytes_with_bit[:: max(1, len(bytes_with_bit) // 80)]:
            b = int(bit_b[by])
            hA, nA = probe(ra, b)
            hB, nB = probe(rb, b)
            if not int(ra[2]) & (0x40000000 if hA else 0x10000000):
                continue
            nfast += 1
            assert (nA if hA else nB if hB else cf + b) == got[n][by], (n, by)
    return nfast, int(img[15])


def decode_want(flat, p):
    """expected next state in the plan's numbering for all (state, byte)"""
    ref = reference_table(flat)
    new2old = p.get("new2old").astype(np.int64)
    new2old[p.S1 - 1] = flat.nstates
    old2new = np.empty(p.S1, np.int64)
    old2new[new2old] = np.arange(p.S1)
    return old2new[ref[new2old]]


def check_plan(flat, layout):
    try:
        p = Plan(flat, layout)
    except OSError:
        return False  # this layout cannot hold this DFA (ENOTSUP)
    ref = reference_table(flat)
    S, S1, Cn = flat.nstates, p.S1, p.C
    cls = p.get("cls").astype(np.int64)
    new2old = p.get("new2old").astype(np.int64)
    new2old[S1 - 1] = S
    old2new = np.empty(S1, np.int64)
    old2new[new2old] = np.arange(S1)
    fin = p.get("fin")
    # fin: original id for end states, NO_MATCH otherwise
    want_fin = np.where(np.append(flat.is_end, 0).astype(bool)[new2old], new2old, NO).astype(np.uint32)
    assert np.array_equal(fin, want_fin)
    assert old2new[flat.start] == p.start
    # eager outputs: masks follow the renumbering, states with outputs sit in the two test ranges
    em = p.get("emask")
    if flat.eager_off is not None:
        eids = p.get("eager_ids")
        assert len(em) == S1 and list(eids) == sorted(set(flat.eager_ids.tolist()))
        for nidx in range(S1 - 1):
            want_ids = flat.eager_of(int(new2old[nidx]))
            if len(eids) <= 64:
                got_ids = eids[[b for b in range(len(eids)) if (int(em[nidx]) >> b) & 1]]
                assert np.array_equal(got_ids, want_ids)
            else:   # wide sets: the mask only flags the state, the ids are in the (word, mask) lists (test_wide_eager_sets)
                assert (em[nidx] != 0) == (len(want_ids) > 0)
            assert (len(want_ids) > 0) == (nidx < p.eager_lo_end or nidx >= p.eager_hi_begin)
        assert em[S1 - 1] == 0
    else:
        assert len(em) == 0
    # expected next state in NEW numbering for all (new state, byte)
    want = old2new[ref[new2old]]            # [S1][256]
    # absorbing threshold
    absorbing = (want == np.arange(S1)[:, None]).all(axis=1)
    assert absorbing[p.abs_min:].all() and not absorbing[:p.abs_min].any()
    assert p.nabsorbing == S1 - p.abs_min
    bytes_ = np.arange(256)
    if p.layout == LAYOUT_TINY:
        col = p.get("tiny_col")
        got = np.stack([(col >> np.uint64(4 * s)) & np.uint64(15) for s in range(S1)]).astype(np.int64)
        col5 = p.get("tiny5_col").astype(np.int64)
        assert (len(col5) == 256) == (S1 <= 6)
        if len(col5):   # Tiny5Pol: state code 5*s, field s of the byte's column = 5 * next(s)
            got5 = np.stack([(col5 >> (5 * s)) & 31 for s in range(S1)])
            assert np.array_equal(got5, 5 * got) and (col5 < (1 << 30)).all()
    elif p.layout in (LAYOUT_LDS, LAYOUT_LDSSELF):
        tab = p.get("lds_tab").astype(np.int64)
        rb = p.row_bytes
        st = np.arange(S1)[:, None] * rb                      # encoded state = byte offset of row
        e = tab[(st + cls[bytes_][None, :] * 2) // 2]
        got = (e << 2) // rb
        assert ((e << 2) % rb == 0).all()
        if p.layout == LAYOUT_LDSSELF:                        # each row ends with the state's self-loop mask
            assert Cn <= 32 and rb == ((Cn + 1) // 2 * 2) * 2 + 4
            sm = tab[(st[:, 0] + rb - 4) // 2] | (tab[(st[:, 0] + rb - 2) // 2] << 16)
            rep = np.array([np.nonzero(cls == c)[0][0] for c in range(Cn)])
            loops = want[:, rep] == np.arange(S1)[:, None]
            bits = (sm[:, None] >> np.arange(Cn)[None, :]) & 1
            assert np.array_equal(bits[~absorbing].astype(bool), loops[~absorbing])
            assert (sm[absorbing] == 0xFFFFFFFF).all()
    elif p.layout == LAYOUT_LDS2:
        # stride 2: T[state][c1 * C1 + c2] = row index (entries) of the state two bytes on; class C is the identity.
        # Every (state, byte) through the single-byte form next() = pair(c, ident); every state x a sample of byte PAIRS
        # (and (ident, byte), (ident, ident)) through the pair form
        tab = p.get("lds_tab").astype(np.int64)
        C1 = Cn + 1
        row = C1 * C1
        assert p.row_bytes == row * 2 and len(tab) == S1 * row and S1 * row <= 61440
        base = np.arange(S1)[:, None] * row
        e = tab[base + cls[bytes_][None, :] * C1 + Cn]
        assert (e % row == 0).all()
        got = e // row
        rng2 = np.random.RandomState(S1)
        b1, b2 = rng2.randint(0, 256, 400), rng2.randint(0, 256, 400)
        e2 = tab[base + (cls[b1] * C1 + cls[b2])[None, :]] // row
        mid = want[:, b1]                                                   # [S1][400]: the state after the first byte
        assert np.array_equal(e2, want[mid, b2[None, :]])
        assert np.array_equal(tab[base[:, 0] + Cn * C1 + Cn] // row, np.arange(S1))
        assert np.array_equal(tab[base + (Cn * C1 + cls[bytes_])[None, :]] // row, want)
    elif p.layout in (LAYOUT_COMB, LAYOUT_COMBSELF):
        comb = p.get("comb").astype(np.int64)
        off = p.get("comb_off").astype(np.int64)
        dfl = p.get("comb_dflt").astype(np.int64)
        cfin = p.get("comb_fin")
        assert len(set(off.tolist())) == S1                    # distinct row offsets
        st = off[:, None]
        x = comb[st + cls[bytes_][None, :]] ^ (st << 16)
        nxt_off = np.where(x < 0x10000, x, dfl[cls[bytes_]][None, :])
        back = np.full(len(comb), -1, np.int64)
        back[off] = np.arange(S1)
        got = back[nxt_off]
        assert np.array_equal(cfin[off], fin)
        assert ((off >= p.comb_abs_min_off) == absorbing).all()
        if len(em):   # eager outputs: the two thresholds hold on row offsets too (DEAD may sit above the upper one)
            emits = (off < p.comb_eager_lo_off) | (off >= p.comb_eager_hi_off)
            assert np.array_equal(emits[:S1 - 1], em[:S1 - 1] != 0)
        if p.layout == LAYOUT_COMBSELF:
            sm = p.get("comb_smask").astype(np.int64)
            assert len(sm) == len(comb) and Cn <= 32
            # bit c of the mask at a state's row offset <=> class c loops to the state itself
            rep = np.array([np.nonzero(cls == c)[0][0] for c in range(Cn)])
            loops = want[:, rep] == np.arange(S1)[:, None]
            bits = (sm[off][:, None] >> np.arange(Cn)[None, :]) & 1
            assert np.array_equal(bits[~absorbing].astype(bool), loops[~absorbing])
            assert (sm[off][absorbing] == 0xFFFFFFFF).all()
            if Cn <= 31:   # the spare class 31 ("no byte": what step16_part gives the bytes beyond an input's end) loops everywhere
                assert ((sm[off] >> 31) & 1).all()
            # self-loop BYTE range for the SWAR chunk test: exactly the state's self-loop bytes, or "none"
            rng = p.get("comb_rng").astype(np.int64)[off]
            lo, hi = rng & 0xff, rng >> 8
            selfb = want == np.arange(S1)[:, None]                       # [S1][256]
            for n in range(S1):
                if rng[n] == 0x0080:
                    continue
                assert lo[n] <= 128 and (hi[n] <= 127 or hi[n] == 255)
                assert np.array_equal(selfb[n], (bytes_ >= lo[n]) & (bytes_ <= hi[n])), (n, lo[n], hi[n])
            assert (rng[absorbing] == 0xFF00).all()
    elif p.layout == LAYOUT_COMB256:
        comb = p.get("comb256").astype(np.int64)
        off = p.get("comb256_off").astype(np.int64)
        cfin = p.get("comb256_fin")
        assert len(set(off.tolist())) == S1
        st = off[:, None]
        e = comb[st + bytes_[None, :]]                         # next << 16 | owner
        nxt_off = np.where((e & 0xffff) == st, e >> 16, p.comb256_dflt)
        back = np.full(len(comb), -1, np.int64)
        back[off] = np.arange(S1)
        got = back[nxt_off]
        assert np.array_equal(cfin[off], fin)
        assert ((off >= p.comb256_abs_min_off) == absorbing).all()
        if len(em):
            emits = (off < p.comb256_eager_lo_off) | (off >= p.comb256_eager_hi_off)
            assert np.array_equal(emits[:S1 - 1], em[:S1 - 1] != 0)
    elif p.layout == LAYOUT_SPARSE:
        got = decode_sparse(p, S1)
        check_sparse_fast(p, got)
    else:
        tab = p.get("glob_tab").astype(np.int64)
        st = np.arange(S1)[:, None] * Cn * 4
        e = tab[(st + cls[bytes_][None, :] * 4) // 4]
        got = e // (Cn * 4)
        # ... and its 2-byte form (Glob16Pol: the next state's index at row * C * 2 + class * 2) where the automaton has <= 65 535 states
        t16 = p.get("glob_tab16").astype(np.int64)
        assert (len(t16) != 0) == (S1 <= 65535)
        if len(t16):
            rank = p.get("glob16_rank").astype(np.int64)          # rows in visit-frequency order (empty: renumbered order)
            if len(rank) == 0:
                rank = np.arange(S1)
            assert np.array_equal(np.sort(rank), np.arange(S1)) and np.array_equal(rank[p.abs_min:], np.arange(p.abs_min, S1))
            assert np.array_equal(t16[rank[:, None] * Cn + cls[bytes_][None, :]], rank[got])
    assert np.array_equal(got, want)
    return True


@pytest.mark.parametrize("path", all_golden_paths(), ids=golden_id)
def test_layouts_encode_delta(path, built):
    flat = Golden(path).flat
    ok = {L: check_plan(flat, L) for L in ALL_LAYOUTS}
    assert ok[LAYOUT_GLOBAL], "the global layout must hold any DFA"
    assert check_plan(flat, 0)


def test_auto_layout_choices(built):
    import os
    from common import GOLDEN
    assert Plan(Golden(os.path.join(GOLDEN, "c1.npz")).flat).layout == LAYOUT_TINY
    c3 = Plan(Golden(os.path.join(GOLDEN, "c3.npz")).flat)
    assert c3.layout in (LAYOUT_COMBSELF, LAYOUT_COMB256, LAYOUT_COMB)   # the ~4k-state union must stay LDS resident
    # a literal set too big for LDS: base-row records (the failure-link form), an order of magnitude smaller
    rng = np.random.RandomState(9)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", np.uint8)
    words = [bytes(alpha[rng.randint(0, 36, rng.randint(6, 12))]) for _ in range(3000)]
    ac = FlatDfa.from_strings(words, 2, list(range(len(words))))
    assert Plan(ac).layout == LAYOUT_GLOBAL          # 3 MB: an L2-resident plain table with a big LDS mirror
    pa = Plan(ac, LAYOUT_SPARSE)
    img = pa.get("sparse")
    assert img.size * 4 * 4 < pa.S1 * pa.C * 4 and int(img[12]) <= 7
    # breadth-first numbering makes every trie node's children consecutive: (nearly) no exception lists remain
    assert int(img[13]) > 0.9 * (int(img[8]) - int(img[10])) * 0.5 and int(img[11]) < 0.05 * pa.S1
    want = decode_want(ac, pa)
    got = decode_sparse(pa, pa.S1)
    assert np.array_equal(got, want)
    # the straight-line walk (SparseFastPol) agrees wherever its own test lets a lane through (here the depth-1 nodes are
    # not full -- not every 2-gram occurs -- so most misses need a fourth level and stay with the general loop)
    nfast, nflag = check_sparse_fast(pa, got)
    assert nfast > 0 and nflag > 0


def test_sparse_fullbase_shortcut(built):
    """A literal set over a small alphabet in which every 2-gram occurs: the depth-1 nodes own all children
    (records with every bit set) and their children are flagged FULLBASE -- a miss resolves as first(base) + bit
    without visiting the base.  The decoder above follows that shortcut; delta must still be exact."""
    rng = np.random.RandomState(17)
    alpha = np.frombuffer(b"abcd", np.uint8)
    words = sorted(set(bytes(alpha[rng.randint(0, 4, rng.randint(3, 9))]) for _ in range(400)))
    ac = FlatDfa.from_strings(words, 2, list(range(len(words))))
    pa = Plan(ac, LAYOUT_SPARSE)
    img = pa.get("sparse")
    # FULLBASE records, CONSEC records; the only exception lists are those of deep records re-based onto shallow ones
    assert int(img[14]) >= 16 and int(img[13]) > 0 and int(img[11]) < int(img[8])
    got = decode_sparse(pa, pa.S1)
    assert np.array_equal(got, decode_want(ac, pa))
    # ... and with full depth-1 nodes every record's misses resolve within two more levels: FASTMISS (nearly) everywhere,
    # deep records included (re-based onto LDS-resident ancestors): the straight-line walk takes (nearly) every byte
    nfast, nflag = check_sparse_fast(pa, got)
    nrec = int(img[8]) - int(img[10])
    assert nflag >= 0.95 * nrec and nfast >= 0.85 * nrec * 4, (nfast, nflag, nrec)


def test_wide_eager_sets(built):
    """More than 64 eager ids: per state the (word, mask) pairs spell exactly the state's id list."""
    from libfsm_amd.capi import RANGE_DTYPE
    rng = np.random.RandomState(3)
    S, nids = 500, 300
    nt = np.full((S, 256), -1, np.int64)
    for c in b"abc":
        nt[:, c] = rng.randint(0, S, S)
    flat = FlatDfa.from_dense(nt, 0, [1] * S)
    pool = np.arange(5, 5 + 3 * nids, 3, dtype=np.uint32)
    off, ids = [0], []
    for s_ in range(S):
        ids.extend(sorted(set(int(x) for x in rng.choice(pool, rng.randint(0, 5)))))
        off.append(len(ids))
    ids.extend(sorted(set(pool.tolist()) - set(ids)))
    ids[off[S - 1]:] = sorted(set(ids[off[S - 1]:]))
    off[-1] = len(ids)
    flat.eager_off, flat.eager_ids = np.array(off, np.uint32), np.array(ids, np.uint32)
    p = Plan(flat, LAYOUT_GLOBAL)
    eids = p.get("eager_ids")
    assert list(eids) == pool.tolist()
    eo, ew, em = p.get("ew_off"), p.get("ew_word"), p.get("ew_mask")
    new2old = p.get("new2old")
    flags = p.get("emask")
    assert len(eo) == p.S1 + 1
    for n in range(p.S1 - 1):
        got = []
        for k in range(int(eo[n]), int(eo[n + 1])):
            got.extend(int(eids[64 * int(ew[k]) + b]) for b in range(64) if (int(em[k]) >> b) & 1)
        want = flat.eager_of(int(new2old[n])).tolist()
        assert got == want
        assert (flags[n] != 0) == (len(want) > 0) == (n < p.eager_lo_end or n >= p.eager_hi_begin)
    assert eo[p.S1 - 1] == eo[p.S1]


def test_rejects_non_dfa(built):
    from libfsm_amd.capi import RANGE_DTYPE
    r = np.zeros(2, RANGE_DTYPE)
    r["lo"], r["hi"], r["to"] = [0, 5], [10, 20], [0, 0]     # overlapping ranges on bytes 5..10
    flat = FlatDfa(1, 0, np.array([0, 2], np.uint32), r, np.array([1], np.uint8), np.zeros(2, np.uint32), np.zeros(0, np.uint32))
    with pytest.raises(OSError):
        Plan(flat)
    r["to"] = [0, 7]
    r["lo"], r["hi"] = [0, 30], [10, 40]
    with pytest.raises(OSError):                               # destination out of range
        Plan(FlatDfa(1, 0, np.array([0, 2], np.uint32), r, np.array([1], np.uint8), np.zeros(2, np.uint32), np.zeros(0, np.uint32)))


def test_random_dfas_all_layouts(built):
    rng = np.random.RandomState(5)
    for S in (1, 2, 7, 16, 17, 300, 2000):
        nt = rng.randint(0, S, (S, 256)).astype(np.int64)
        nt[rng.rand(S, 256) < 0.5] = -1
        ncls = rng.randint(1, 40)
        colmap = rng.randint(0, ncls, 256)
        nt = nt[:, colmap]                                     # force few byte classes
        flat = FlatDfa.from_dense(nt, int(rng.randint(S)), rng.randint(0, 2, S))
        for L in (0,) + tuple(ALL_LAYOUTS):
            check_plan(flat, L)
        # the same automaton with eager outputs on a third of its states (narrow: <= 64 ids, wide: more):
        # every layout must keep "emits" a two-threshold test on its state encoding
        for nids in (40, 200):
            off, ids = [0], []
            for s_ in range(S):
                if rng.rand() < 0.33:
                    ids.extend(sorted(set(int(x) for x in rng.randint(1, nids + 1, rng.randint(1, 4)))))
                off.append(len(ids))
            if not ids:
                continue
            flat.eager_off, flat.eager_ids = np.array(off, np.uint32), np.array(ids, np.uint32)
            for L in (0,) + tuple(ALL_LAYOUTS):
                check_plan(flat, L)


def test_layouts_on_reference_fsm_corpus(built):
    """Every layout that can hold each of the 319 corpus automata encodes its transition function exactly."""
    from common import Corpus
    c = Corpus()
    held = 0
    for k in range(0, len(c)):
        flat = c.get(k)[0]
        for L in (0,) + tuple(ALL_LAYOUTS):
            held += bool(check_plan(flat, L))
    assert held > 4 * len(c)


# ---- the lazy form of the sparse layout (plan.cpp build_lazy, walk_lazy.h) ---------------------------------

SENT = 0x80000000


def lazy_decode(p):
    img = p.get("lazy").astype(np.int64)
    if img.size == 0:
        return None
    assert img[0] == 0x31595a4c
    H, F, fwords, rec_off, filt_off, lds_bytes, grec_off = (int(x) for x in img[1:8])
    assert filt_off == 1024 and rec_off == filt_off + 4 * fwords and lds_bytes == rec_off + 16 * (H + 1) <= 160 * 1024
    assert fwords & (fwords - 1) == 0 and int(img[13]) == p.S1
    L = img[16:16 + lds_bytes // 4]
    sh = L[:256]
    filt = L[filt_off // 4: filt_off // 4 + fwords]
    rec = L[rec_off // 4:].reshape(H + 1, 4)
    ncl = int(img[12])                                   # clones: ids S1 .. S1 + ncl - 1 (round 6), own records behind the states', cloneof[] behind car[]
    own = img[grec_off // 4: grec_off // 4 + 4 * (p.S1 + ncl)].reshape(p.S1 + ncl, 4)
    car = img[int(img[14]) // 4: int(img[14]) // 4 + p.S1]
    cloneof = img[int(img[14]) // 4 + p.S1: int(img[14]) // 4 + p.S1 + ncl]
    assert int(img[14]) // 4 == grec_off // 4 + 4 * (p.S1 + ncl)
    for j in range(ncl):                                 # a clone stands for a real, non-absorbing state beyond the LDS set and has its record
        t = int(cloneof[j])
        assert H <= t < p.abs_min and np.array_equal(own[p.S1 + j], own[t]), (j, t)
    return dict(H=H, F=F, fwords=fwords, sh=sh, filt=filt, rec=rec, own=own, car=car, abs_reach=int(img[11]), nkeys=int(img[8]), nz=int(img[9]), nsent=int(img[10]),
                S1=p.S1, ncl=ncl, cloneof=cloneof)


def lazy_probe(b0, b1, fm1, sh):
    """walk_lazy.h lazy_probe: (hit, first - 1 + rank + 1) of the class's bit in {bits, fm1}"""
    x = (((b1 << 32) | b0) << (sh & 63)) & 0xFFFFFFFFFFFFFFFF
    return (x >> 63) & 1 == 1, (bin(x).count("1") + fm1) & 0xFFFFFFFF


def lazy_step(z, abs_min, sid, E, sh, use_abs):
    """one byte of walk_lazy.h lazy_step on state (id, E): returns (m, rep, sentinel); the own record is consulted
    exactly when the kernel's `pos` says so (ids >= F, or the filter bit of (id, sh % 32))"""
    H, F = z["H"], z["F"]
    rb = z["rec"][E] if E <= H else np.zeros(4, np.int64)
    hB, nB = lazy_probe(int(rb[0]), int(rb[1]), int(rb[2]), sh)
    cfb = (int(rb[3]) - sh) & 0xFFFFFFFF
    ev = nB if hB else cfb
    s32 = lambda v: v - (1 << 32) if v & SENT else v
    evD = s32(ev) >= H
    rep = cfb if evD else ev
    pos = sid >= F or (sid >= H and (int(z["filt"][sid & (z["fwords"] - 1)]) >> (sh & 31)) & 1)
    m = ev
    if pos and sid < len(z["own"]):
        g = z["own"][sid]
        hA, pc = lazy_probe(int(g[0]), int(g[1]), 0, sh)      # {bits, base, stride}: the k-th exception leads to base + k * stride
        if hA:
            stride = int(g[3]) - (1 << 32) if int(g[3]) & SENT else int(g[3])
            m = (int(g[2]) + pc * stride) & 0xFFFFFFFF
    if use_abs and abs_min <= sid < z["S1"]:             # (clone ids lie beyond the absorbing range: the kernel's range test)
        m, rep = sid, E
    return m, rep, bool((m | rep | sh) & SENT)


def check_lazy(p, want, max_pairs=None):
    """Every reachable (state, carried record) pair of the lazy walk, every byte: the kernel's step -- sentinel -> the exact
    path (here: the dense table), after which the state carries what the planner's car[] says -- lands on delta(state, byte).  A state may be reached
    with several carried records only if its steps do not depend on them (the planner then makes its own record except
    everything); the pairs are followed one by one, so that is checked too."""
    z = lazy_decode(p)
    if z is None:
        return None
    H, N = z["H"], p.abs_min
    assert p.start < H
    use_abs = bool(z["abs_reach"])
    seen = {(p.start, p.start)}
    todo = [(p.start, p.start)]
    nfast = nsent = nsent_bit = 0
    real = lambda x: int(z["cloneof"][x - p.S1]) if x >= p.S1 else x     # the state an id stands for (walk_lazy.h lazy_unclone)
    while todo:
        s, E = todo.pop()
        if real(s) >= N:
            continue
        for by in range(256):
            sh = int(z["sh"][by])
            m, rep, sent = lazy_step(z, N, s, E, sh, use_abs)
            t = int(want[real(s)][by])
            if sent:
                nsent += 1
                nsent_bit += not (sh & SENT)
                m, rep = t, (t if t < H else int(z["car"][t]))            # the exact path; a state beyond the LDS set is handed what it carries
            else:
                nfast += 1
                assert m < p.S1 + z["ncl"] and real(m) == t, (s, E, by, m, t)
                assert rep <= H
                if m >= p.S1:                                # a clone is entered with what the state it stands for carries (or that state carries
                    assert rep == int(z["car"][t]) or int(z["car"][t]) == H, (s, by, m, rep)   # nothing -- Z: its record excepts everything)
            if real(m) >= N:
                assert use_abs, "an absorbing state is reachable: the kernel variant that tests for it must be the one launched"
                continue
            if m < H:
                assert rep == m, (s, by, m, rep)
            if (m, rep) not in seen:
                seen.add((m, rep))
                todo.append((m, rep))
        if max_pairs and len(seen) > max_pairs:
            break
    states = len({s for s, _ in seen})
    return dict(H=H, F=z["F"], states=states, pairs=len(seen), fast=nfast, sentinel=nsent, sentinel_bit=nsent_bit, nkeys=z["nkeys"], nz=z["nz"])


def test_lazy_form_of_literal_sets(built):
    """The lazy walk's image for literal sets of several shapes: full depth-1 / depth-2 levels (the shape of configs[4]),
    sparse ones, anchored and unanchored: exact on every reachable (state, carried, byte)."""
    rng = np.random.RandomState(23)
    shapes = [(b"abcd", 400, 3, 9, 2), (b"abcdefgh", 1500, 4, 10, 2), (b"abcdefghijklmnop", 2000, 5, 9, 0),
              (b"abcdefghijklmnopqrstuvwxyz0123456789", 1500, 6, 12, 2), (b"ab", 60, 2, 8, 2)]
    seen_lazy = 0
    for alpha_b, nw, lo, hi, flags in shapes:
        alpha = np.frombuffer(alpha_b, np.uint8)
        words = sorted(set(bytes(alpha[rng.randint(0, len(alpha), rng.randint(lo, hi + 1))]) for _ in range(nw)))
        ac = FlatDfa.from_strings(words, flags, list(range(len(words))))
        pa = Plan(ac, LAYOUT_SPARSE, lds_limit=64 * 1024 if len(alpha_b) <= 8 else 0)
        want = decode_want(ac, pa)
        r = check_lazy(pa, want)
        if r is None:
            continue
        seen_lazy += 1
        assert r["states"] >= r["H"]
        # where the first levels are full the straight-line path takes (nearly) every byte
        # (sentinel_bit: sentinel steps on bytes of the alphabet; the others take the exact path by design)
        print(alpha_b, r)
    assert seen_lazy >= 3


def test_lazy_form_on_random_dfas(built):
    """Literal-set automata with a few per cent of their transitions rewired at random (still DFAs, no longer tries: states
    entered in several ways, exceptions that are no progression, targets inside the LDS set ...): whatever the automaton,
    the lazy image -- when the planner makes one -- is exact on every reachable (state, carried record, byte)."""
    rng = np.random.RandomState(5)
    made = 0
    for trial in range(8):
        alpha_b = [b"abcd", b"abcdefgh", b"abcdefghijkl"][trial % 3]
        alpha = np.frombuffer(alpha_b, np.uint8)
        words = sorted(set(bytes(alpha[rng.randint(0, len(alpha), rng.randint(3, 8))]) for _ in range(int(rng.randint(200, 1500)))))
        ac = FlatDfa.from_strings(words, int(rng.choice([0, 2])), list(range(len(words))))
        nt = ac.dense().astype(np.int64)
        nt[nt == NO] = -1
        S = ac.nstates
        k = int(S * len(alpha_b) * rng.choice([0.002, 0.02, 0.1]))
        ss, cc = rng.randint(0, S, k), alpha[rng.randint(0, len(alpha), k)]
        nt[ss, cc] = rng.randint(0, S, k)
        if trial % 4 == 3:
            nt[rng.randint(0, S, 5)] = -1                      # a few states without edges: DEAD becomes reachable
        flat = FlatDfa.from_dense(nt, ac.start, ac.is_end)
        pa = Plan(flat, LAYOUT_SPARSE, lds_limit=int(rng.choice([48, 160])) * 1024)
        r = check_lazy(pa, decode_want(flat, pa))
        made += r is not None
    assert made >= 4


def test_workgroup_size_by_occupancy(built):
    """fsm_hip_waves_by_occupancy: the rule that sizes the workgroup of the latency-bound per-lane kernels (walk_lines32) from
    the kernel's register count -- round 5: two 12-wavefront workgroups per CU where one of 16 fitted were worth 20 % on short
    lines.  Checked against a brute-force restatement of the hardware limits (512 registers per SIMD lane in steps of 8, 8
    wavefronts per SIMD, 32 per CU, the table's LDS)."""
    import libfsm_amd as hip

    def brute(vg, by_lds, wmax):
        per_simd = min(8, 512 // ((vg + 7) // 8 * 8))
        best, best_res = wmax, 0
        for w in range(wmax, 7, -4):
            blocks = min(per_simd // ((w + 3) // 4), by_lds, 32 // w)
            if blocks * w > best_res:
                best, best_res = w, blocks * w
        return best

    assert hip.waves_by_occupancy(76, 2, 16) == 12      # walk_lines32<CombSelfPol> beside the C3 table: 2 x 12, not 1 x 16
    assert hip.waves_by_occupancy(80, 2, 16) == 12
    assert hip.waves_by_occupancy(88, 2, 16) == 16      # 5 wavefronts per SIMD: one workgroup either way, the larger one
    assert hip.waves_by_occupancy(64, 2, 16) == 16      # 8 per SIMD: two of 16
    assert hip.waves_by_occupancy(64, 1, 16) == 16      # the table's LDS allows one
    assert hip.waves_by_occupancy(40, 8, 16) == 16
    assert hip.waves_by_occupancy(0, 2, 16) == 16       # unknown register count: what the caller had
    assert hip.waves_by_occupancy(76, 2, 4) == 4        # small workgroups are left alone
    for vg in range(8, 257, 4):
        for by_lds in (1, 2, 3, 4, 8):
            for wmax in (8, 12, 16):
                assert hip.waves_by_occupancy(vg, by_lds, wmax) == brute(vg, by_lds, wmax), (vg, by_lds, wmax)
