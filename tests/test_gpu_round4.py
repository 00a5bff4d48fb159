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


def test_packed_all_every_metadata_form(hip):
    """fsm_hip_exec_batch_packed_all{,_device}: end states, accept bitmap, device-side end-ids (EARLIEST and RET) and eager
    sets from ONE walk over packed lines whose metadata is u64 offsets, u32 offsets or lengths alone -- the (b, e) lines the
    generated matchers take (print/c.c:569-619), ids included.  Short lines (the per-lane kernel) and long ones (the
    lane-refilling kernel); end-ids against the oracle's fsm_endid_get sets on the C3 automaton, eager sets against the
    golden ids of tests/eager_output programs; host and device pointers; every form must give the same answers."""
    import torch
    from common import eager_golden_paths
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c3.npz"))
    o = Oracle(g.flat)
    rng = np.random.RandomState(41)
    a = np.frombuffer(b"abcdwxyz0123456789", np.uint8)
    pats = bytes(np.load(os.path.join(GOLDEN, "c3.npz"))["patterns"]).split(b"\n")

    def line(k):
        if rng.randint(3):
            return bytes(a[rng.randint(0, len(a), k)])
        p = pats[rng.randint(len(pats))]
        return p[1:p.index(b"[")] + bytes(rng.randint(48, 58, max(1, k - 6)).astype(np.uint8)) + b"yz"

    def metas(strings):
        lens = np.array([len(s) for s in strings], np.uint32)
        off = np.zeros(len(strings) + 1, np.uint64)
        off[1:] = np.cumsum(lens)
        base = np.frombuffer(b"".join(strings) + b"\0", np.uint8)
        return base, ((hip.META_OFF64, off), (hip.META_OFF32, off.astype(np.uint32)), (hip.META_LENGTHS, lens))

    dfa = hip.HipDfa(g.flat)
    sets = dfa.ret_sets()
    for lo, hi, n in ((0, 60, 6001), (0, 700, 3000), (5, 6, 1), (0, 1, 130)):
        strings = [line(rng.randint(lo, hi)) for _ in range(n)]
        ret, want = o.exec_strings(strings)
        base, forms = metas(strings)
        for mode in (1, 2):
            for form, meta in forms:
                r = dfa.exec_packed_all_form(base, form, meta, n, ids_mode=mode, want_bitmap=True)
                assert np.array_equal(r["end"], want), (lo, hi, mode, form)
                assert np.array_equal(np.unpackbits(r["bitmap"].view(np.uint8), bitorder="little")[:n].astype(bool), want != NO)
                ids = r["ids"]
                assert (ids[want == NO] == NO).all()
                for i in np.nonzero(want != NO)[0][:300]:
                    e = o.endids(int(want[i]))
                    if mode == 1:
                        assert ids[i] == (int(e[0]) if len(e) else 0xFFFFFFFE)
                    else:
                        assert np.array_equal(sets[ids[i]], e)
                # device pointers, ids only (no end states, no bitmap)
                d_base = torch.from_numpy(base.copy()).cuda()
                d_meta = torch.from_numpy(meta.view(np.int64) if form == hip.META_OFF64 else meta.view(np.int32)).cuda()
                d_ids = torch.full((n,), 7, dtype=torch.int32, device="cuda")
                dfa.exec_packed_all_device(d_base.data_ptr(), form, d_meta.data_ptr(), n, ids_mode=mode, d_ids=d_ids.data_ptr())
                torch.cuda.synchronize()
                assert np.array_equal(d_ids.cpu().numpy().view(np.uint32), ids), (lo, hi, mode, form)
    dfa.close()
    checked = 0
    for path in eager_golden_paths():
        ge = Golden(path)
        strs = ge.strings()
        strs = strs + [s * 9 for s in strs]                        # long lines too
        eo = Oracle(ge.flat)
        d = hip.HipDfa(ge.flat)
        rows = np.zeros((len(strs), max(1, max(len(s) for s in strs))), np.uint8)
        lens = np.array([len(s) for s in strs], np.uint32)
        for i, s in enumerate(strs):
            rows[i, :len(s)] = np.frombuffer(s, np.uint8)
        _, wend, wsets = eo.exec_eager(rows, lens)
        base, forms = metas(strs)
        k = d.eager_id_count()
        idv = np.array([d.eager_id(b) for b in range(k)], np.uint32)
        for form, meta in forms:
            r = d.exec_packed_all_form(base, form, meta, len(strs), want_eager=True)
            assert np.array_equal(r["end"], np.where(wend == NO, NO, wend)), (path, form)
            bits = np.unpackbits(r["eager"].view(np.uint8).reshape(len(strs), -1), axis=1, bitorder="little")[:, :k].astype(bool)
            for i in range(len(strs)):
                assert np.array_equal(idv[bits[i]], wsets[i]), (path, form, i)
            checked += len(strs)
        d.close()
    assert checked >= 500
    # error contracts: a bad form, NULL metadata, decreasing offsets
    dfa = hip.HipDfa(g.flat)
    with pytest.raises(OSError):
        dfa.exec_packed_all_form(np.zeros(4, np.uint8), 7, np.zeros(2, np.uint32), 1)
    with pytest.raises(OSError):
        dfa.exec_packed_all_form(np.zeros(8, np.uint8), hip.META_OFF32, np.array([4, 2, 8], np.uint32), 2)
    dfa.close()


def test_device_fronts_capture_into_a_hip_graph(hip):
    """The device-pointer fronts allocate nothing and synchronise nothing once a dfa is warm, so a serving loop can capture
    them into a HIP graph (small batches are launch-bound: 13 us per replay against 23 us per launch at 4 096 inputs):
    every front captured on torch's capture stream, the graph replayed on CHANGED inputs, against the oracle."""
    import torch
    from oracle.pyoracle import Oracle
    for name in ("c3.npz", "c1.npz"):
        g_ = Golden(os.path.join(GOLDEN, name))
        o = Oracle(g_.flat)
        dfa = hip.HipDfa(g_.flat)
        rng = np.random.RandomState(3)
        n, L = 4096 + 5, 256
        alpha = np.frombuffer(b"abcdwxyzLlibfsm0123456789", np.uint8)

        def batch():
            rows = alpha[rng.randint(0, len(alpha), (n, L))]
            lens = rng.randint(0, L + 1, n).astype(np.uint32)
            off = np.zeros(n + 1, np.uint64)
            off[1:] = np.cumsum(lens)
            packed = np.concatenate([rows[i, :lens[i]] for i in range(n)] + [np.zeros(16, np.uint8)])
            return rows, lens, off, packed

        rows, lens, off, packed = batch()
        d_rows = torch.from_numpy(rows).cuda()
        d_len = torch.from_numpy(lens.view(np.int32)).cuda()
        d_off = torch.from_numpy(off.view(np.int64)).cuda()
        d_packed = torch.zeros(n * L + 16, dtype=torch.uint8, device="cuda")
        d_packed[:len(packed)] = torch.from_numpy(packed).cuda()
        d_end = torch.zeros(n, dtype=torch.int32, device="cuda")
        d_bm = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
        fronts = {
            "stride": (lambda s: dfa.exec_batch_device(d_rows.data_ptr(), L, n, d_end.data_ptr(), d_bm.data_ptr(), stream=s), False),
            "stride+len": (lambda s: dfa.exec_batch_device(d_rows.data_ptr(), L, n, d_end.data_ptr(), d_bm.data_ptr(), d_len=d_len.data_ptr(), stream=s), True),
            "offsets": (lambda s: dfa.exec_batch_offsets_device(d_packed.data_ptr(), d_off.data_ptr(), n, d_end.data_ptr(), d_bm.data_ptr(), stream=s), True),
            "lengths": (lambda s: dfa.exec_batch_lengths_device(d_packed.data_ptr(), d_len.data_ptr(), n, d_end.data_ptr(), d_bm.data_ptr(), stream=s), True),
        }
        graphs = {}
        for fname, (call, _) in fronts.items():
            call(torch.cuda.current_stream().cuda_stream)      # warm: lazily built tables, scratch blocks
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                call(torch.cuda.current_stream().cuda_stream)
            graphs[fname] = gr
        for rep in range(3):                                    # new bytes, lengths and offsets in the SAME device buffers
            rows, lens, off, packed = batch()
            d_rows.copy_(torch.from_numpy(rows))
            d_len.copy_(torch.from_numpy(lens.view(np.int32)))
            d_off.copy_(torch.from_numpy(off.view(np.int64)))
            d_packed[:len(packed)] = torch.from_numpy(packed).cuda()
            for fname, (_, ragged) in fronts.items():
                want = o.table_walk(rows, lens if ragged else None)
                d_end.fill_(7)
                d_bm.fill_(-1)
                graphs[fname].replay()
                torch.cuda.synchronize()
                assert np.array_equal(d_end.cpu().numpy().view(np.uint32), want), (name, fname, rep)
                bits = np.unpackbits(d_bm.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)
                assert np.array_equal(bits, want != NO), (name, fname, rep, int((bits & ~(want != NO)).sum()), int((~bits & (want != NO)).sum()), dfa.last_kernel_name())
        dfa.close()


def test_lazy_walk_captures_into_a_hip_graph(hip):
    """The lazy walk claims its tiles from a device counter that every launch zeroes first (a kernel node): captured once,
    replayed on three different row sets, against the oracle."""
    import torch
    from oracle.pyoracle import Oracle
    rng = np.random.RandomState(99)
    alpha_b = b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_-"
    words, flat = literal_set(hip, rng, alpha_b, 20000, 8, 16, 2)
    o = Oracle(flat)
    dfa = hip.HipDfa(flat, hip.LAYOUT_SPARSE)
    dfa.tune(KNOB_SPARSE_FAST, 3)
    n, L = 6000 + 17, 256
    d_rows = torch.zeros((n, L), dtype=torch.uint8, device="cuda")
    d_end = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_bm = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    call = lambda s: dfa.exec_batch_device(d_rows.data_ptr(), L, n, d_end.data_ptr(), d_bm.data_ptr(), stream=s)   # noqa: E731
    call(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert "walk_lazy" in dfa.last_kernel_name()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        call(torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        rows = rows_over(rng, alpha_b, n, L, words, every=3, foreign=0.001)
        want = o.table_walk(rows)
        d_rows.copy_(torch.from_numpy(rows))
        d_end.fill_(7)
        d_bm.fill_(-1)
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        assert np.array_equal(d_end.cpu().numpy().view(np.uint32), want), rep
        bits = np.unpackbits(d_bm.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)
        assert np.array_equal(bits, want != NO), rep
    dfa.close()
