"""GPU (-m gpu): parity tests added in round 2, all through the C ABI.

  * the eager-output kernels the round-1 suite never reached: LDS-DMA and per-lane-load kernels on
    uniform 128-byte-multiple rows, the ragged kernel, and eager outputs in the comb / sparse layouts;
  * a > 6 GB resident batch: rows at byte offsets >= 2^32, the persistent grid's tail and the last partial
    tile, pulled back and compared with the oracle;
  * the ragged (coalesced, lane-refilling) kernel on hostile length distributions;
  * ret-list order for end-ids >= 256 (cmp_ret is a memcmp), AMBIG_ERROR, chunked match_file,
    two host threads sharing one dfa, the multi-device front, retest -l hip.
Bit-exact everywhere."""
import ctypes
import os
import sys
import subprocess
import threading

import numpy as np
import pytest

from common import GOLDEN, Golden, eager_golden_paths

pytestmark = pytest.mark.gpu

NO = 0xFFFFFFFF
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    torch.cuda.set_device(0)
    import libfsm_amd
    libfsm_amd.load_library()
    return libfsm_amd


def bits(bm, n):
    return np.unpackbits(bm.view(np.uint8), bitorder="little")[:n].astype(bool)


def _missing_integration(what):
    """A missing integration binary: skip -- or fail when FSM_REQUIRE_INTEGRATION is set (a GPU box that is expected to
    carry the prebuilt files: without this a clean checkout would quietly drop the SURVEY 8(a10) tests)."""
    if os.environ.get("FSM_REQUIRE_INTEGRATION"):
        pytest.fail(what + " and FSM_REQUIRE_INTEGRATION is set")
    pytest.skip(what)


def _need_ref():
    from oracle import pyoracle
    if not pyoracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")


EAGER_LAYOUTS = ("TINY", "COMBSELF", "COMB256", "LDSSELF", "LDS", "COMB", "SPARSE", "GLOBAL", "AUTO")


def _eager_modes(hip):
    return ((hip.IN_LDSDMA, 0), (hip.IN_LDSDMA, 5), (hip.IN_DIRECT, 0), (hip.IN_RAGGED, 0), (hip.IN_GENERIC, 0), (-1, 0))


def test_eager_outputs_fast_kernels_golden_dfas(hip):
    """Every tests/eager_output automaton on uniform 256-byte rows (no length array), so that the walk takes
    walk_ldsdma<EagerPol<...>> / walk_direct<EagerPol<...>> -- the kernels behind the eager throughput numbers --
    and the ragged and generic ones on the same rows: ids and end states against the oracle."""
    from oracle.pyoracle import Oracle
    rng = np.random.RandomState(77)
    n_checked = n_fired = 0
    for path in eager_golden_paths():
        g = Golden(path)
        pats = [p.encode("latin1").strip(b"^$") for p in g.meta["patterns"]]
        alpha = np.frombuffer((" ".join(g.meta["patterns"]) + " xyz").encode("latin1"), np.uint8)
        rows = alpha[rng.randint(0, len(alpha), (330, 256))]
        for i in range(0, 330, 3):                  # plant whole patterns so that outputs fire
            p = pats[rng.randint(len(pats))]
            if 0 < len(p) <= 60 and not any(c in p for c in b"[]()*+?|\\."):
                at = rng.randint(0, 256 - len(p))
                rows[i, at:at + len(p)] = np.frombuffer(p, np.uint8)
        wret, wend, wsets = Oracle(g.flat).exec_eager(rows)
        n_fired += sum(len(s) for s in wsets)
        for lname in EAGER_LAYOUTS:
            try:
                dfa = hip.HipDfa(g.flat, getattr(hip, "LAYOUT_" + lname))
            except OSError:
                continue
            for mode, waves in _eager_modes(hip):
                dfa.tune(hip.KNOB_INPUT_MODE, mode)
                dfa.tune(hip.KNOB_WAVES, waves)
                end, sets = dfa.exec_batch_eager(rows)
                assert np.array_equal(end, wend), (g.meta["source"], lname, mode, waves)
                for i in range(len(rows)):
                    assert np.array_equal(sets[i], wsets[i]), (g.meta["source"], lname, mode, waves, i)
                n_checked += len(rows)
            dfa.close()
    assert n_checked > 100000 and n_fired > 2000


def test_eager_64bit_columns_repeated_launches(hip):
    """Regression / tripwire: eager walks of the 7..16-state goldens in the TINY layout, every input mode at
    4 / 8 / 12 wavefronts, 40 launches each on two alternating row sets (so a stale result is a wrong one).
    One build of the ragged kernel failed ~3 % of these launches (walk_kernels.h, note in TinyPol::next);
    the single launch per configuration of the test above would mostly miss that."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import eager_tiny64_stress
    lines = []
    n, bad = eager_tiny64_stress.run(40, out=lines.append)
    assert n >= 4 * 3 * 40 and bad == 0, lines
    # and the ragged kernel on packed inputs, plain walks, every layout of the C1 / C3 automata: 30 launches each
    lines = []
    n, bad = eager_tiny64_stress.run_packed(30, out=lines.append)
    assert n >= 300 and bad == 0, lines


@pytest.mark.parametrize("nids", [40, 64, 65, 300])
def test_eager_outputs_fast_kernels_random_dfas(hip, nids):
    """Random DFAs with register-held (<= 64 ids) and wide id sets: uniform rows through the LDS-DMA (register
    sets; wide sets fall back to per-lane loads inside launch_walk), direct, ragged and generic kernels, in every
    layout that can hold the automaton -- including comb / combself / comb256 / sparse, new in round 2."""
    from oracle.pyoracle import Oracle
    from test_gpu_parity import random_eager_dfa
    rng = np.random.RandomState(1000 + nids)
    for S in (6, 13, 250, 3000, 30000):
        flat = random_eager_dfa(rng, S, nids)
        rows = np.frombuffer(b"abcdefgh", np.uint8)[rng.randint(0, 8, (513, 128))]
        lens = rng.randint(0, 129, 513).astype(np.uint32)
        o = Oracle(flat)
        want = {None: o.exec_eager(rows, None, cap=nids + 8), "r": o.exec_eager(rows, lens, cap=nids + 8)}
        held = 0
        for lname in EAGER_LAYOUTS:
            try:
                dfa = hip.HipDfa(flat, getattr(hip, "LAYOUT_" + lname))
            except OSError:
                continue
            held += 1
            assert dfa.eager_id_count() == nids
            for mode, waves in _eager_modes(hip):
                dfa.tune(hip.KNOB_INPUT_MODE, mode)
                dfa.tune(hip.KNOB_WAVES, waves)
                for key, ln in ((None, None), ("r", lens)):
                    end, sets = dfa.exec_batch_eager(rows, ln)
                    _, wend, wsets = want[key]
                    assert np.array_equal(end, wend), (S, lname, mode, waves, key)
                    for i in range(len(rows)):
                        assert np.array_equal(sets[i], wsets[i]), (S, lname, mode, waves, key, i)
            dfa.close()
        assert held >= 3


def test_eager_union_in_a_comb_layout_live_reference(hip):
    """A start-anchored eager union built by the reference (fsm_union_repeated_pattern_group) that lands in a
    comb layout (AUTO picks combself for it): before round 2 the planner sent every eager DFA that was not
    tiny/lds to `global`."""
    _need_ref()
    from oracle.pyoracle import RefFsm
    rng = np.random.RandomState(4)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", np.uint8)
    # anchored at the start only: a pattern anchored at both ends yields an end-id, not an eager output
    pats = sorted(set(b"^" + bytes(alpha[rng.randint(0, 26, 3)]) + b"[0-9]+(x|yz)" for _ in range(48)))[:40]
    f = RefFsm.union_repeated("pcre", pats, 1, False)
    strings = []
    for i in range(900):
        p = pats[rng.randint(len(pats))]
        s = p[1:4] + bytes(rng.randint(48, 58, rng.randint(1, 60)).astype(np.uint8)) + (b"x" if i % 2 else b"yz")
        strings.append(s if i % 3 else bytes(alpha[rng.randint(0, 26, rng.randint(0, 30))]))
    ret, end, sets = f.exec_eager_strings(strings)
    stride = 128
    rows = np.zeros((len(strings), stride), np.uint8)
    lens = np.array([len(s) for s in strings], np.uint32)
    for i, s in enumerate(strings):
        rows[i, :len(s)] = np.frombuffer(s, np.uint8)
    layouts = set()
    for L in (hip.LAYOUT_AUTO, hip.LAYOUT_COMBSELF, hip.LAYOUT_COMB256, hip.LAYOUT_COMB, hip.LAYOUT_SPARSE):
        try:
            dfa = hip.HipDfa.compile_fsm(f.ptr, L)
        except OSError:
            continue
        layouts.add(dfa.info()["layout_name"])
        gend, gsets = dfa.exec_batch_eager(rows, lens)
        assert np.array_equal(gend, end), L
        for i in range(len(strings)):
            assert np.array_equal(gsets[i], sets[i]), (L, strings[i])
        dfa.close()
    assert layouts & {"combself", "comb256", "comb"}, layouts
    assert sum(len(s) for s in sets) > 300


# ---------------------------------------------------------------------------
# beyond 4 GiB
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("workload", ["c2", "c3"])
def test_rows_beyond_4GiB_and_the_last_partial_tile(hip, workload):
    """6.3e6 x 1 KiB = 6.45 GB resident: rows around byte offset 2^32 (row 4 194 304), a seeded spread over the
    whole range, the rows walked last by the persistent grid and the last partial tile (n is not a multiple of
    64) are pulled back and compared with the oracle -- for the default kernel, per-lane loads and the ragged
    kernel.  The device rows themselves are compared with the host generator (counter-based: any row can be
    regenerated on its own)."""
    import torch
    import bench
    from oracle.pyoracle import Oracle
    n, L = 6_300_037, 1024
    flat = hip.FlatDfa.load(os.path.join(GOLDEN, "c1.npz" if workload == "c2" else "c3.npz"))
    buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
    bench.generate(hip, workload, buf.data_ptr(), n, L, 0)
    torch.cuda.synchronize()
    edge = 1 << 22                                       # row whose first byte sits at offset 2^32
    idx = np.unique(np.concatenate([
        np.arange(edge - 160, edge + 160), np.arange(n - 300, n), np.arange(0, 130),
        np.arange(2 * edge - 70, 2 * edge + 70) if 2 * edge + 70 < n else np.zeros(0, np.int64),
        np.random.RandomState(5).randint(0, n, 6000)])).astype(np.int64)
    rows = buf[torch.from_numpy(idx).cuda()].cpu().numpy()
    # the generator twin, on contiguous runs of the sample
    for lo, cnt in ((edge - 160, 320), (n - 300, 300)):
        assert np.array_equal(buf[lo:lo + cnt].cpu().numpy(), bench.generate_host(hip, workload, cnt, L, lo))
    want = Oracle(flat).table_walk(rows)
    assert 0 < int((want != NO).sum()) < len(idx)
    end = torch.empty(n, dtype=torch.int32, device="cuda")
    bm = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    dfa = hip.HipDfa(flat)
    ref_end = None
    for mode in (-1, hip.IN_DIRECT, hip.IN_RAGGED):
        dfa.tune(hip.KNOB_INPUT_MODE, mode)
        end.fill_(-2)
        bm.fill_(-1)
        dfa.exec_batch_device(buf.data_ptr(), L, n, end.data_ptr(), bm.data_ptr())
        torch.cuda.synchronize()
        got = end[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), (workload, mode)
        assert int((end == -2).sum()) == 0                                   # every row was written
        b = bits(bm.cpu().numpy(), n)
        assert np.array_equal(b[idx], want != NO)
        assert int(b.sum()) == int((end != -1).sum())
        tail = bm.cpu().numpy().view(np.uint64)[-1]
        assert int(tail) >> (n % 64) == 0                                    # no bits beyond n in the last word
        if ref_end is None:
            ref_end = end.clone()
        else:
            assert torch.equal(ref_end, end), (workload, mode)               # all 6.3e6 rows agree across kernels
    dfa.close()


# ---------------------------------------------------------------------------
# the ragged kernel on hostile length distributions
# ---------------------------------------------------------------------------

def _packed(strings):
    off = np.zeros(len(strings) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in strings])
    return np.frombuffer(b"".join(strings), np.uint8), off


@pytest.mark.parametrize("name,alpha", [("c1.npz", b"Llibfsmx\0"), ("c3.npz", b"abcdwxyz0123456789")])
def test_ragged_kernel_length_distributions(hip, name, alpha):
    """Packed inputs whose lengths are (a) uniform 0..1024, (b) mostly empty, (c) a few very long ones among
    short ones (a lane keeps one input for hundreds of segments while its neighbours are refilled), (d) all
    exactly 128 / 127 / 129 bytes, (e) one input: ragged (default) and generic kernels, 1 to 12 waves, against
    the oracle -- end states, bitmap, and through the ids / resume fronts."""
    from oracle.pyoracle import Oracle
    rng = np.random.RandomState(len(alpha))
    a = np.frombuffer(alpha, np.uint8)
    g = Golden(os.path.join(GOLDEN, name))
    o = Oracle(g.flat)

    def rnd(k):
        return bytes(a[rng.randint(0, len(a), k)])

    pats = None
    if name == "c3.npz":
        pats = bytes(np.load(os.path.join(GOLDEN, "c3.npz"))["patterns"]).split(b"\n")

    def accepted(k):
        if pats is None:
            s = bytearray(rnd(max(k, 6)))
            at = rng.randint(0, len(s) - 5)
            s[at:at + 6] = b"Libfsm"
            return bytes(s)
        p = pats[rng.randint(len(pats))]
        return p[1:p.index(b"[")] + bytes(rng.randint(48, 58, max(1, k - 6)).astype(np.uint8)) + b"yz"

    cases = {
        "uniform": [accepted(rng.randint(8, 1025)) if i % 4 == 0 else rnd(rng.randint(0, 1025)) for i in range(9000)],
        "mostly_empty": [b"" if i % 7 else rnd(rng.randint(0, 300)) for i in range(5000)],
        "long_among_short": [accepted(100_000) if i % 257 == 3 else rnd(rng.randint(0, 40)) for i in range(3000)],
        "exact128": [accepted(128) if i % 2 else rnd(128) for i in range(1500)],
        "exact127": [rnd(127) for _ in range(700)],
        "exact129": [rnd(129) for _ in range(700)],
        "single": [accepted(5000)],
        "all_empty": [b""] * 300,
    }
    for cname, strings in cases.items():
        ret, want = o.exec_strings(strings)
        base, off = _packed(strings)
        for L in (hip.LAYOUT_AUTO, hip.LAYOUT_LDS, hip.LAYOUT_GLOBAL, hip.LAYOUT_COMB256):
            try:
                dfa = hip.HipDfa(g.flat, L)
            except OSError:
                continue
            for mode, waves, align in ((hip.IN_RAGGED, 0, 0), (hip.IN_RAGGED, 1, 0), (hip.IN_RAGGED, 7, 0), (hip.IN_GENERIC, 0, 0)):
                dfa.tune(hip.KNOB_INPUT_MODE, mode)
                dfa.tune(hip.KNOB_WAVES, waves)
                for early in (1, 0):
                    dfa.tune(hip.KNOB_EARLY_RETIRE, early)
                    end, bm = dfa.exec_batch_offsets(base, off)
                    assert np.array_equal(end, want), (name, cname, L, mode, waves, align, early)
                    assert np.array_equal(bits(bm, len(strings)), ret == 1), (name, cname, L, mode, waves, align, early)
            dfa.close()
    # stride + lengths through the ragged kernel, with the id and resume fronts
    rows = a[rng.randint(0, len(a), (4000, 272))]
    lens = rng.randint(0, 273, 4000).astype(np.uint32)
    ret, want = o.exec_stride(rows, lens)
    dfa = hip.HipDfa(g.flat)
    dfa.tune(hip.KNOB_INPUT_MODE, hip.IN_RAGGED)
    end, bm = dfa.exec_batch(rows, lens)
    assert np.array_equal(end, want)
    st, end2 = dfa.exec_batch_resume(rows, np.full(len(rows), hip.STATE_START, np.uint32), lens)
    assert np.array_equal(st, o.state_walk(rows, np.full(len(rows), hip.STATE_START, np.uint32), lens))
    assert np.array_equal(end2, want)
    ids = dfa.exec_batch_ids(rows, 1, lens)
    for i in np.nonzero(want != NO)[0][:200]:
        e = o.endids(int(want[i]))
        assert ids[i] == (int(e[0]) if len(e) else 0xFFFFFFFE)
    assert (ids[want == NO] == NO).all()
    dfa.close()


def test_ragged_kernel_batch_end_and_tiny_inputs(hip):
    """The ragged kernel fetches 16-byte pieces from the inputs' own byte addresses and never reads beyond the
    batch: an input's last partial piece comes from (end - 16), and inputs shorter than 16 bytes that sit within
    16 bytes of the batch's end are assembled from byte loads.  Device-resident batches of EXACTLY the inputs'
    size (packed, and fixed stride < 16 with lengths): every total from 0 to 40 bytes, tiny inputs at the end
    after long ones, all lengths 0..33 at every start alignment -- against the oracle."""
    import torch
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c1.npz"))        # [Ll]ibf+(sm)*, unanchored: "libf" anywhere accepts
    o = Oracle(g.flat)
    rng = np.random.RandomState(11)
    a = np.frombuffer(b"Llibfsmx\0", np.uint8)

    def rnd(k):
        return bytes(a[rng.randint(0, len(a), k)])

    def tiny(k):
        s = bytearray(rnd(k))
        if k >= 4 and rng.randint(2):
            at = rng.randint(0, k - 3)
            s[at:at + 4] = b"libf"
        return bytes(s)

    dfa = hip.HipDfa(g.flat)
    dfa.tune(hip.KNOB_INPUT_MODE, hip.IN_RAGGED)
    batches = []
    for total in range(0, 41):                                   # whole batch smaller than / around one piece
        cut = sorted(rng.randint(0, total + 1, rng.randint(0, 6)))
        body = tiny(total)
        batches.append([body[x:y] for x, y in zip([0] + cut, cut + [total])])
    for k in range(60):                                          # long inputs, then tiny ones up to the last byte
        batches.append([tiny(rng.randint(0, 700)) for _ in range(rng.randint(1, 150))] + [tiny(rng.randint(0, 16)) for _ in range(rng.randint(1, 9))])
    batches.append([tiny(L) for L in range(34) for _ in range(16)])           # every length at every alignment
    batches.append([tiny(L) for L in range(33, -1, -1) for _ in range(17)])
    for strings in batches:
        ret, want = o.exec_strings(strings)
        base, off = _packed(strings)
        d_base = torch.from_numpy(base.copy()).cuda() if len(base) else torch.empty(0, dtype=torch.uint8, device="cuda")
        d_off = torch.from_numpy(off.view(np.int64)).cuda()
        d_end = torch.full((len(strings),), 7, dtype=torch.int32, device="cuda")
        d_bm = torch.zeros((len(strings) + 63) // 64, dtype=torch.int64, device="cuda")
        for waves in (0, 1):
            dfa.tune(hip.KNOB_WAVES, waves)
            dfa.exec_batch_offsets_device(d_base.data_ptr(), d_off.data_ptr(), len(strings), d_end.data_ptr(), d_bm.data_ptr())
            torch.cuda.synchronize()
            assert np.array_equal(d_end.cpu().numpy().view(np.uint32), want), (len(strings), len(base))
            assert np.array_equal(bits(d_bm.cpu().numpy(), len(strings)), ret == 1)
    # fixed stride below 16 with lengths: the rows at the end of the buffer take the byte-load path
    for stride in (1, 3, 5, 15, 16, 17, 31):
        n = 1000 + stride
        rows = np.stack([np.frombuffer(tiny(stride), np.uint8) for _ in range(n)])
        lens = rng.randint(0, stride + 1, n).astype(np.uint32)
        ret, want = o.exec_stride(rows, lens)
        d_rows = torch.from_numpy(rows).cuda()
        d_len = torch.from_numpy(lens.view(np.int32)).cuda()
        d_end = torch.full((n,), 7, dtype=torch.int32, device="cuda")
        dfa.exec_batch_device(d_rows.data_ptr(), stride, n, d_end.data_ptr(), 0, d_len=d_len.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d_end.cpu().numpy().view(np.uint32), want), stride
    dfa.close()


def test_ragged_kernel_large_packed_batch_on_device(hip):
    """2e6 packed inputs of 0..1024 bytes resident on the device (1 GB): the ragged kernel agrees with
    walk_generic on every input, a seeded sample agrees with the oracle, popcount(bitmap) = #accepts."""
    import torch
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c1.npz"))
    n, L = 2_000_000, 1024
    rng = np.random.RandomState(3)
    lens = rng.randint(0, L + 1, n).astype(np.int64)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    total = int(off[-1])
    buf = torch.empty(((total + 1023) // 1024 + 1, 1024), dtype=torch.uint8, device="cuda")
    hip.gen_inputs_device(buf.data_ptr(), buf.shape[0], 1024, 0, 0x5EEDF5A1, None, b"Libfsm", 2)
    d_off = torch.from_numpy(off.view(np.int64)).cuda()
    e = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(2)]
    bm = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    dfa = hip.HipDfa(g.flat)
    for k, mode in enumerate((hip.IN_RAGGED, hip.IN_GENERIC)):
        dfa.tune(hip.KNOB_INPUT_MODE, mode)
        dfa.exec_batch_offsets_device(buf.data_ptr(), d_off.data_ptr(), n, e[k].data_ptr(), bm.data_ptr() if k == 0 else 0)
    torch.cuda.synchronize()
    assert torch.equal(e[0], e[1])
    assert int((e[0] != -1).sum()) == int(bits(bm.cpu().numpy(), n).sum()) > n // 8
    host = buf.reshape(-1)[:total].cpu().numpy()
    idx = rng.randint(0, n, 5000)
    strings = [bytes(host[int(off[i]):int(off[i + 1])]) for i in idx]
    ret, want = Oracle(g.flat).exec_strings(strings)
    assert np.array_equal(e[0].cpu().numpy().view(np.uint32)[idx], want)
    dfa.close()


# ---------------------------------------------------------------------------
# end-id delivery: ret order, AMBIG_ERROR
# ---------------------------------------------------------------------------

def test_ret_order_is_cmp_ret_and_ambig_error(hip):
    """The ret list is ordered like build_retlist sorts it (src/libfsm/vm/retlist.c:63-79: by count, then
    memcmp over the uint32 ids -- byte-wise, so {256} sorts BEFORE {1} on a little-endian host), and
    FSM_HIP_IDS_ERROR refuses a DFA with an end state that carries two ids (print/c.c:67-72)."""
    S = 9
    nt = np.full((S, 256), -1, np.int64)
    for k in range(1, S):
        nt[0, ord("a") + k - 1] = k
    idsets = [[], [1], [256], [0x01000000], [2, 3], [2, 0x100], [70000], [1], []]
    off, ids = [0], []
    for s_ in range(S):
        ids.extend(idsets[s_])
        off.append(len(ids))
    flat = hip.FlatDfa.from_dense(nt, 0, [0] + [1] * (S - 1))
    flat.endid_off, flat.endids = np.array(off, np.uint32), np.array(ids, np.uint32)
    dfa = hip.HipDfa(flat)
    sets = [tuple(int(x) for x in s) for s in dfa.ret_sets()]
    key = lambda t: (len(t), np.array(t, "<u4").tobytes())
    want = sorted(set(tuple(x) for x in idsets[1:]), key=key)
    assert sets == want
    assert sets.index((256,)) < sets.index((1,)) and sets.index((0x01000000,)) < sets.index((1,))   # not numeric
    rows = np.zeros((S - 1, 16), np.uint8)
    lens = np.ones(S - 1, np.uint32)
    rows[:, 0] = np.arange(S - 1) + ord("a")
    r = dfa.exec_batch_ids(rows, 2, lens)
    for k in range(1, S):
        assert sets[int(r[k - 1])] == tuple(idsets[k])
    e = dfa.exec_batch_ids(rows, 1, lens)
    assert list(e) == [(min(s) if s else 0xFFFFFFFE) for s in idsets[1:]]
    assert dfa.ids_conflict() == 4                      # lowest end state with two ids
    with pytest.raises(OSError):
        dfa.exec_batch_ids(rows, 3, lens)
    dfa.close()
    # without a conflict AMBIG_ERROR is AMBIG_EARLIEST
    flat2 = hip.FlatDfa.from_dense(nt, 0, [0] + [1] * (S - 1))
    flat2.endid_off = np.array([0] + list(range(0, S)), np.uint32)        # state 0: none; states 1..S-1: one id each
    flat2.endids = np.array([300 - k for k in range(S - 1)], np.uint32)
    d2 = hip.HipDfa(flat2)
    assert d2.ids_conflict() is None
    assert np.array_equal(d2.exec_batch_ids(rows, 3, lens), d2.exec_batch_ids(rows, 1, lens))
    assert list(d2.exec_batch_ids(rows, 3, lens)) == [300 - k for k in range(S - 1)]
    d2.close()


# ---------------------------------------------------------------------------
# streaming file front, threads
# ---------------------------------------------------------------------------

def test_match_file_in_chunks(hip, tmp_path):
    """fsm_hip_match_file carries the state across 64 KiB chunks (fsm_vm_match_file's loop, vm.c:188-216):
    files below, at and far above the chunk size, the match straddling a chunk boundary, the empty file; an
    anchored DFA that dies on the first byte stops reading at once."""
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c1.npz"))
    o = Oracle(g.flat)
    dfa = hip.HipDfa(g.flat)
    rng = np.random.RandomState(6)
    cases = [b"", b"Libfsm", b"x" * 65536, b"x" * 65533 + b"Libfsm" + b"y" * 10, b"x" * 200_000 + b"libffsm",
             bytes(rng.randint(0, 256, 300_000).astype(np.uint8)), b"q" * 131072, b"Libf" + b"x" * 70000]
    for k, data in enumerate(cases):
        p = tmp_path / f"f{k}.bin"
        p.write_bytes(data)
        ret, _ = o.exec_one(data)
        assert dfa.match_file(str(p)) == ret == dfa.match_buffer(data), k
    assert dfa.state_is_absorbing(hip.STATE_DEAD) and not dfa.state_is_absorbing(hip.STATE_START)
    dfa.close()
    g3 = Golden(os.path.join(GOLDEN, "c3.npz"))
    d3 = hip.HipDfa(g3.flat)
    p = tmp_path / "big.bin"
    p.write_bytes(b"#" + b"0" * 3_000_000)               # '#' has no edge from the start state: DEAD after one byte
    assert d3.match_file(str(p)) == 0
    pats = bytes(np.load(os.path.join(GOLDEN, "c3.npz"))["patterns"]).split(b"\n")
    good = pats[5][1:pats[5].index(b"[")] + b"7" * 150_000 + b"x"
    p.write_bytes(good)
    assert d3.match_file(str(p)) == 1 == Oracle(g3.flat).exec_one(good)[0]
    d3.close()


def test_two_host_threads_share_one_dfa(hip):
    """The per-dfa mutex: two threads hammering the host-pointer front of ONE dfa (ctypes drops the GIL) get the
    right answers; a third uses the device front on its own stream meanwhile."""
    import torch
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c3.npz"))
    o = Oracle(g.flat)
    dfa = hip.HipDfa(g.flat)
    rng = np.random.RandomState(11)
    a = np.frombuffer(b"abcdwxyz0123456789", np.uint8)
    jobs = []
    for t in range(2):
        rows = a[rng.randint(0, len(a), (3000 + 500 * t, 96))]
        lens = rng.randint(0, 97, len(rows)).astype(np.uint32)
        jobs.append((rows, lens, o.exec_stride(rows, lens)[1]))
    drows = torch.from_numpy(g.rows).cuda()
    dend = torch.empty(len(g.rows), dtype=torch.int32, device="cuda")
    errors = []

    def host(rows, lens, want):
        try:
            for _ in range(60):
                end, _bm = dfa.exec_batch(rows, lens)
                if not np.array_equal(end, want):
                    errors.append("host mismatch")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def device():
        try:
            s = torch.cuda.Stream()
            for _ in range(60):
                dfa.exec_batch_device(drows.data_ptr(), drows.shape[1], len(g.rows), dend.data_ptr(), 0, stream=s.cuda_stream)
                s.synchronize()
                if not np.array_equal(dend.cpu().numpy().view(np.uint32), g.end):
                    errors.append("device mismatch")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=host, args=j) for j in jobs] + [threading.Thread(target=device)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]
    dfa.close()


# ---------------------------------------------------------------------------
# retest -l hip: the reference's own retest(1) with IMPL_HIP in its runner
# ---------------------------------------------------------------------------

def test_retest_l_hip(hip, tmp_path):
    """integration/retest: the reference's retest main.c + runner.c with the IMPL_HIP patch applied
    (integration/retest/impl_hip.patch; built by integration/retest/build.sh against the reference archive).
    `retest -l hip` over the reference's tests/retest cases (37 regexps / 115 +/- lines, re-emitted in .tst
    format from the frozen goldens): one fsm_hip_compile per regexp -- in the forked child, so the HIP context
    is created after the fork.  `-l hip` (round 5) reads the whole file ahead and matches every record's lines in ONE
    fsm_hip_exec_multi; `-l hip-record` reads ahead to the end of each record and matches its test lines in
    one launch (fsm_hip_exec_batch_offsets; a [BATCH ] line per record says how many), `-l hip-line` keeps one
    fsm_hip_match_buffer launch per test line.  0 errors; the same file through `-l vm` (the reference's own
    interpreter) prints the same [OK] lines; a flipped expectation is reported by all three.  The read-ahead's
    control flow is also covered without a GPU in tests/test_retest_patch.py."""
    from common import retest_tst_lines
    exe = os.path.join(ROOT, "integration", "_build", "retest")
    if not os.path.exists(exe):
        sh = subprocess.run(["sh", os.path.join(ROOT, "integration", "retest", "build.sh")], capture_output=True, text=True)
        if not os.path.exists(exe):
            _missing_integration("integration/_build/retest not built (needs /root/reference at build time): " + sh.stderr[-300:])
    lines, flip = retest_tst_lines()
    tst = tmp_path / "all.tst"
    tst.write_bytes(("\n".join(lines) + "\n").encode("latin1"))
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    oks = {}
    for impl in ("hip", "hip-record", "hip-line", "vm"):
        out = subprocess.run([exe, "-l", impl, str(tst)], capture_output=True, text=True, errors="replace", env=env, timeout=600)
        tail = out.stdout.strip().splitlines()[-2:]
        assert out.returncode == 0, (impl, out.stdout[-1500:], out.stderr[-1500:])
        assert tail[0].endswith("37 regexps, 115 test cases") and tail[1].endswith("0 re errors, 0 errors"), (impl, tail)
        assert out.stdout.count("[OK    ]") == 115 and "[NOT OK]" not in out.stdout
        oks[impl] = [l for l in out.stdout.splitlines() if l.startswith("[OK")]
        batch = [l for l in out.stdout.splitlines() if l.startswith("[BATCH ]")]
        if impl == "hip":      # round 5: the whole file in ONE fsm_hip_exec_multi -- one copy in, ONE kernel, one copy out
            assert len(batch) == 1 and "37 records, 115 test lines matched in 1 launch," in batch[0], batch
        else:                  # a record per launch (hip-record), a launch per line, the reference's VM
            held = [int(l.split(":")[1].split()[0]) for l in batch]
            assert sum(held) == (115 if impl == "hip-record" else 0), (impl, held)
    assert oks["hip"] == oks["vm"] == oks["hip-line"] == oks["hip-record"]
    lines[flip] = "-" + lines[flip][1:]
    bad = tmp_path / "bad.tst"
    bad.write_bytes(("\n".join(lines) + "\n").encode("latin1"))
    for impl in ("hip", "hip-record", "hip-line", "vm"):
        out = subprocess.run([exe, "-l", impl, str(bad)], capture_output=True, text=True, errors="replace", env=env, timeout=600)
        assert out.returncode == 1 and out.stdout.count("[NOT OK]") == 1, (impl, out.stdout[-800:])
        assert out.stdout.strip().splitlines()[-1].endswith("0 re errors, 1 errors")


def test_reperf_l_hip(hip, tmp_path):
    """The reference's reperf(1) with the same runner patch (integration/_build/reperf): reperf/boost.scr's five string
    cases, re-emitted from the frozen goldens.  `-l hip` sends the N runs of a case as batches of N copies
    (fsm_runner_run_repeat -> fsm_hip_exec_batch: 10^6 runs = one launch), `-l hip-line` makes N calls as the loop at
    src/retest/reperf.c:772-784 does; both agree with `-l vm`, and an inverted expectation is reported by all."""
    from common import reperf_scr_lines
    exe = os.path.join(ROOT, "integration", "_build", "reperf")
    if not os.path.exists(exe):
        _missing_integration("integration/_build/reperf not built (needs /root/reference at build time)")
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))

    def go(impl, lines):
        scr = tmp_path / "t.scr"
        scr.write_text("\n".join(lines) + "\n")
        return subprocess.run([exe, "-C", "-l", impl, str(scr)], capture_output=True, text=True, errors="replace", env=env, timeout=600)

    strip = lambda t: [l for l in t.splitlines() if "iterations took" not in l]
    N = 1_000_000                                     # the script's own N
    out = go("hip", reperf_scr_lines(N))
    assert out.returncode == 0, (out.stdout[-800:], out.stderr[-800:])
    assert out.stdout.count("execute %d iterations took" % N) == 5
    ref = go("vm", reperf_scr_lines(N))
    assert strip(out.stdout) == strip(ref.stdout)
    line = go("hip-line", reperf_scr_lines(300))
    assert line.returncode == 0 and line.stdout.count("execute 300 iterations took") == 5
    bad, badref = go("hip", reperf_scr_lines(5000, flip=3)), go("vm", reperf_scr_lines(5000, flip=3))
    assert strip(bad.stdout) == strip(badref.stdout) and bad.returncode == badref.returncode
    assert strip(bad.stdout) != strip(go("hip", reperf_scr_lines(5000)).stdout)


def test_reference_test_programs_on_the_hip_path(hip):
    """SURVEY section 8(b), "reference C tests re-linked against the shim": the reference's own tests/endids/*.c (16),
    tests/re_strings/*.c (4) and tests/eager_output/*.c (22), compiled where they lie with fsm_exec() routed to
    fsm_hip_compile + fsm_hip_exec -- or, for an automaton with an eager-output callback, fsm_hip_exec_batch_eager and
    one callback call per id of the returned set (integration/reftests).  Their own assert()s -- accept / reject,
    end-id sets after union / determinise / minimise / trim, per-word ids of the Aho-Corasick builds, eager ids of
    anchored / unanchored pattern unions -- all hold with the GPU doing the matching: exit status 0, every call on the
    HIP path, none falling back."""
    import re as _re
    from test_retest_patch import reference_test_programs
    progs = reference_test_programs()
    if len(progs) != 42:
        _missing_integration("integration/_build/reftests not built (needs /root/reference at build time)")
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    total = eager = 0
    for exe in progs:
        out = subprocess.run([exe], capture_output=True, text=True, errors="replace", env=env, timeout=600)
        assert out.returncode == 0, (os.path.basename(exe), out.stdout[-400:], out.stderr[-400:])
        m = _re.search(r"exec_via_hip: (\d+) fsm_exec calls answered by the HIP path, (\d+) fallbacks \((\d+) with eager", out.stderr)
        if m is None:          # a program that builds and inspects automata without executing them (endids6)
            continue
        assert int(m.group(1)) > 0 and int(m.group(2)) == 0, (os.path.basename(exe), out.stderr[-300:])
        # (an eager_output program whose patterns are all anchored carries its ids as end-ids: no eager call)
        assert int(m.group(3)) == 0 or os.path.basename(exe).startswith("eager_"), (os.path.basename(exe), out.stderr[-300:])
        total += int(m.group(1))
        eager += int(m.group(3))
    assert total > 700 and eager > 50


def test_re_H(hip, tmp_path):
    """The reference's re(1) with integration/re/hip_exec.patch (integration/_build/re): `re -H` matches all its text
    arguments in one launch and files (-x) through fsm_hip_match_file; exit status and -z output equal plain `re`'s
    for every case of tests/test_retest_patch.py::RE_CASES, plus a 3 MB file that needs the chunked walk."""
    from test_retest_patch import RE_CASES
    exe = os.path.join(ROOT, "integration", "_build", "re")
    if not os.path.exists(exe):
        _missing_integration("integration/_build/re not built (needs /root/reference at build time)")
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    for args, rc in RE_CASES:
        ref = subprocess.run([exe] + args, capture_output=True, text=True, env=env, timeout=120)
        got = subprocess.run([exe, "-H"] + args, capture_output=True, text=True, env=env, timeout=120)
        assert ref.returncode == rc and got.returncode == rc, (args, got.stderr[-400:])
        assert got.stdout == ref.stdout, args
    f1, f2, f3 = tmp_path / "a.txt", tmp_path / "b.txt", tmp_path / "c.txt"
    f1.write_bytes(b"a" + b"b" * 3_000_000 + b"c")
    f2.write_bytes(b"a" + b"b" * 3_000_000 + b"d")
    f3.write_bytes(b"")
    for files in ([f1], [f2], [f1, f2], [f3]):
        args = ["-r", "pcre", "-x", "^ab+c$", "--"] + [str(f) for f in files]
        ref = subprocess.run([exe] + args, capture_output=True, text=True, env=env, timeout=120)
        got = subprocess.run([exe, "-H"] + args, capture_output=True, text=True, env=env, timeout=120)
        assert ref.returncode == got.returncode, (files, got.stderr[-300:])


def test_fsm_H(hip, tmp_path):
    """The reference's fsm(1) with integration/fsm/hip_exec.patch: `fsm -H file.fsm text...` exits like plain `fsm`
    (tests/test_retest_patch.py::FSM_CASES), -x reads a file through fsm_hip_match_file, an NFA is refused (EINVAL)."""
    from test_retest_patch import FSM_CASES, FSM_DFA, FSM_NFA
    exe = os.path.join(ROOT, "integration", "_build", "fsm")
    if not os.path.exists(exe):
        _missing_integration("integration/_build/fsm not built (needs /root/reference at build time)")
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    dfa, nfa, txt = tmp_path / "d.fsm", tmp_path / "n.fsm", tmp_path / "t.txt"
    dfa.write_text(FSM_DFA)
    nfa.write_text(FSM_NFA)
    txt.write_bytes(b"a" + b"b" * 200_000 + b"cc")
    for texts, rc in FSM_CASES:
        ref = subprocess.run([exe, str(dfa)] + texts, capture_output=True, text=True, env=env, timeout=120)
        got = subprocess.run([exe, "-H", str(dfa)] + texts, capture_output=True, text=True, env=env, timeout=120)
        assert ref.returncode == rc == got.returncode and got.stdout == ref.stdout, (texts, got.stderr[-300:])
    # -x: fsm(1) takes its first remaining argument as the file to read (main.c:741)
    ref = subprocess.run([exe, "-x", str(dfa), str(txt)], capture_output=True, text=True, env=env, timeout=120)
    got = subprocess.run([exe, "-x", "-H", str(dfa), str(txt)], capture_output=True, text=True, env=env, timeout=120)
    assert ref.returncode == got.returncode == 0, got.stderr[-300:]
    got = subprocess.run([exe, "-H", str(nfa), "a"], capture_output=True, text=True, env=env, timeout=120)
    assert got.returncode != 0 and "fsm_hip_compile: Invalid argument" in got.stderr


# ---------------------------------------------------------------------------
# multi-device front (C ABI): one replica per device, one host thread per device
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], "all"])
def test_node_front_shards_and_gathers(hip, devices):
    """fsm_hip_node_*: the batch is split into contiguous shards of whole bitmap words, one per replica, each
    driven by its own host thread.  This box has one GPU, so [0] exercises the RCCL path (a communicator of one:
    ncclAllGather / ncclAllReduce are really called) and [0, 0] / [0, 0, 0] put several replicas on it and take
    the peer-copy exchange.  Host-pointer fronts (stride + lengths, packed) and the device-resident front with
    the whole-batch bitmap and match count on every replica: all against the oracle."""
    import torch
    import bench
    from oracle.pyoracle import Oracle
    if devices == "all":      # every GPU the box shows: RCCL with more than one rank, the moment such a box runs this
        if torch.cuda.device_count() < 2:
            pytest.skip("one GPU")
        devices = list(range(torch.cuda.device_count()))
    g = Golden(os.path.join(GOLDEN, "c3.npz"))
    o = Oracle(g.flat)
    node = hip.HipNode(g.flat, devices)
    assert node.ndev == len(devices)
    assert node.uses_rccl() == (len(set(devices)) == len(devices))
    rng = np.random.RandomState(len(devices))
    # shards: contiguous, whole words, cover [0, n)
    for n in (1, 63, 64, 65, 1000, 100_003):
        cover = 0
        for k in range(node.ndev):
            f, c = node.shard(n, k)
            assert f == cover and (f % 64 == 0 or c == 0)
            cover += c
        assert cover == n
    # host pointers, ragged rows
    a = np.frombuffer(b"abcdwxyz0123456789", np.uint8)
    rows = a[rng.randint(0, len(a), (10_007, 80))]
    lens = rng.randint(0, 81, len(rows)).astype(np.uint32)
    ret, want = o.exec_stride(rows, lens)
    end, bm = node.exec_batch(rows, lens)
    assert np.array_equal(end, want) and np.array_equal(bits(bm, len(rows)), ret == 1)
    pats = bytes(np.load(os.path.join(GOLDEN, "c3.npz"))["patterns"]).split(b"\n")
    strings = [(pats[rng.randint(len(pats))][1:4] + b"123yz") if i % 3 == 0 else bytes(a[rng.randint(0, len(a), rng.randint(0, 50))]) for i in range(5003)]
    ret, want = o.exec_strings(strings)
    end, bm = node.exec_strings(strings)
    assert np.array_equal(end, want) and np.array_equal(bits(bm, len(strings)), ret == 1)
    assert (ret == 1).sum() > 300
    # device-resident shards: each replica generates its own rows by global index, walks them; every replica
    # ends up with the whole bitmap; the count is the batch's
    n, L = 300_037, 1024
    bufs, ends, bms = [], [], []
    W = node.bitmap_words(n)
    for k in range(node.ndev):
        f, c = node.shard(n, k)
        dev = f"cuda:{devices[k]}"
        torch.cuda.set_device(devices[k])
        b = torch.empty((max(c, 1), L), dtype=torch.uint8, device=dev)
        if c:
            bench.generate(hip, "c3", b.data_ptr(), c, L, f)
        bufs.append(b)
        ends.append(torch.full((max(c, 1),), -2, dtype=torch.int32, device=dev))
        bms.append(torch.full((W,), -1, dtype=torch.int64, device=dev))
    for dv in set(devices):
        torch.cuda.synchronize(dv)
    torch.cuda.set_device(0)
    cnt = node.exec_batch_device([b.data_ptr() for b in bufs], L, n, [e.data_ptr() for e in ends], [m.data_ptr() for m in bms], want_count=True)
    host = bench.generate_host(hip, "c3", n, L, 0)
    want = o.table_walk(host)
    got = np.concatenate([ends[k][:node.shard(n, k)[1]].cpu().numpy().view(np.uint32) for k in range(node.ndev)])
    assert np.array_equal(got, want)
    assert cnt == int((want != NO).sum()) > n // 3
    for k in range(node.ndev):
        assert np.array_equal(bits(bms[k].cpu().numpy(), n), want != NO), k            # the whole bitmap on every replica
        assert not bits(bms[k].cpu().numpy(), W * 64)[n:].any()
    # replicas answer end-id queries like a single dfa
    r0 = node.replica(0)
    e = int(want[want != NO][0])
    assert np.array_equal(r0.endids(e), o.endids(e))
    node.close()


def test_bench_node_front_mode(hip):
    """bench.py --node-front: the C multi-device front measured the way bench.py runs it for itself when N = 1 sees
    several GPUs -- here with two replicas on the one GPU (peer-copy exchange) and with the single device (RCCL)."""
    import json
    import sys
    for env_extra, ndev, rccl in (({"FSM_BENCH_NODE_REPLICAS": "2"}, 2, False), ({}, 1, True)):
        env = dict(os.environ, **env_extra)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--node-front", "--n", "200000", "--steps", "2", "--warmup", "1"],
                             capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert len(r["devices"]) == ndev and r["uses_rccl"] == rccl and r["workload"] == "c3"
        assert r["inputs_total"] == 200000 * ndev // (64 * ndev) * (64 * ndev)
        assert r["accepted_inputs"] == r["inputs_total"] // 2 and r["bitmap_popcount_matches_count_on_every_checked_replica"]
        assert r["value_GBps"] > 0
