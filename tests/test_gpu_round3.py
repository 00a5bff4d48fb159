"""GPU (-m gpu): parity tests added in round 3, all through the C ABI.

  * the packed-offsets front (written for walk_packed, the round-3 kernel that was removed in round 4; the cases now
    run walk_generic, walk_ragged and the device-side choice between them, and -- round 4 -- the u32-offsets and
    lengths-only forms of the same batches): every length 0..40 at every alignment, retest-like line sets, hostile
    length mixes, every table layout, offsets arrays at 8-mod-16 addresses, off[0] != 0, batches that end exactly
    on a line;
  against the oracle, bit-exact."""
import os

import numpy as np
import pytest

from common import GOLDEN, Golden

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


def _packed(strings):
    off = np.zeros(len(strings) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in strings])
    return np.frombuffer(b"".join(strings), np.uint8), off


def _layouts(hip, flat):
    out = []
    for L in (hip.LAYOUT_AUTO, hip.LAYOUT_TINY, hip.LAYOUT_LDS, hip.LAYOUT_LDSSELF, hip.LAYOUT_COMB, hip.LAYOUT_COMB256,
              hip.LAYOUT_COMBSELF, hip.LAYOUT_GLOBAL, hip.LAYOUT_SPARSE):
        try:
            out.append((L, hip.HipDfa(flat, L)))
        except OSError:
            pass
    return out


def _cases(name, rng):
    a = np.frombuffer(b"Llibfsmx\0" if name == "c1.npz" else b"abcdwxyz0123456789", np.uint8)
    pats = None
    if name == "c3.npz":
        pats = bytes(np.load(os.path.join(GOLDEN, "c3.npz"))["patterns"]).split(b"\n")

    def rnd(k):
        return bytes(a[rng.randint(0, len(a), k)])

    def accepted(k):
        if pats is None:
            s = bytearray(rnd(max(k, 6)))
            at = rng.randint(0, len(s) - 5)
            s[at:at + 6] = b"Libfsm"
            return bytes(s)
        p = pats[rng.randint(len(pats))]
        return p[1:p.index(b"[")] + bytes(rng.randint(48, 58, max(1, k - 6)).astype(np.uint8)) + b"yz"

    def mix(k):
        return accepted(k) if rng.randint(3) == 0 else rnd(k)

    return {
        "len1to40": [mix(L) for L in range(1, 41) for _ in range(37)],                      # every length at every alignment
        "len40to0": [mix(L) for L in range(40, -1, -1) for _ in range(19)],
        "short8to64": [mix(rng.randint(8, 65)) for _ in range(20000)],
        "tiny0to3": [rnd(rng.randint(0, 4)) for _ in range(9000)],                          # > 6 ends per 16-byte chunk: several passes
        "one_byte": [rnd(1) for _ in range(5000)],
        "mostly_empty": [b"" if i % 7 else mix(rng.randint(0, 90)) for i in range(6000)],
        "all_empty": [b""] * 700,
        "empties_at_both_ends": [b""] * 70 + [mix(rng.randint(1, 50)) for _ in range(500)] + [b""] * 70,
        "long_among_short": [accepted(30_000) if i % 301 == 3 else mix(rng.randint(0, 40)) for i in range(3000)],
        "uniform0to1024": [mix(rng.randint(0, 1025)) for _ in range(3000)],
        "exact16": [mix(16) for _ in range(3000)],
        "exact128": [mix(128) for _ in range(800)],
        "exact127_129": [mix(127 + 2 * (i & 1)) for i in range(800)],
        "single_short": [accepted(9)],
        "single_long": [accepted(5000)],
        "two": [b"", accepted(700)],
    }


@pytest.mark.parametrize("name", ["c1.npz", "c3.npz"])
def test_packed_kernel_length_distributions(hip, name):
    """walk_generic / walk_ragged forced and the auto choice, every layout the DFA can take, 1 to 12 waves; u64 offsets,
    u32 offsets and lengths only: end states and bitmap against the oracle."""
    from oracle.pyoracle import Oracle
    rng = np.random.RandomState(5 + len(name))
    g = Golden(os.path.join(GOLDEN, name))
    o = Oracle(g.flat)
    cases = _cases(name, rng)
    dfas = _layouts(hip, g.flat)
    assert len(dfas) >= 4
    for cname, strings in cases.items():
        ret, want = o.exec_strings(strings)
        base, off = _packed(strings)
        for L, dfa in dfas:
            lens = np.diff(off.astype(np.int64)).astype(np.uint32)
            # (round 5: a plain walk of a packed batch below 4 GiB takes walk_lines32, the 32-bit form of walk_generic; early = 33
            # -- bit 5 of the knob -- keeps it on walk_generic's own body, which batches of 4 GiB and more still run)
            for mode, waves, early in ((hip.IN_GENERIC, 0, -1), (hip.IN_GENERIC, 0, 33), (hip.IN_GENERIC, 1, -1), (hip.IN_RAGGED, 5, -1), (hip.IN_RAGGED, 0, -1), (-1, 0, -1)):
                if L != hip.LAYOUT_AUTO and waves not in (0, 1):
                    continue
                print(name, cname, L, mode, waves, early, flush=True)
                dfa.tune(hip.KNOB_INPUT_MODE, mode)
                dfa.tune(hip.KNOB_WAVES, waves)
                dfa.tune(hip.KNOB_EARLY_RETIRE, early)
                end, bm = dfa.exec_batch_offsets(base, off)
                bad = np.nonzero(end != want)[0]
                assert len(bad) == 0, (name, cname, L, mode, waves, early, len(bad), bad[:8], [len(strings[i]) for i in bad[:8]])
                if mode == hip.IN_GENERIC and len(strings):
                    kn = dfa.last_kernel_name()
                    other = "walk_lazy" in kn or "SparsePol" in kn      # (the record walk keeps walk_generic: launch.h lines32_ok)
                    assert other or ("walk_lines32" in kn) == (early < 0), (kn, early)
                assert np.array_equal(bits(bm, len(strings)), ret == 1), (name, cname, L, mode, waves)
                end, bm = dfa.exec_batch_offsets(base, off, want_bitmap=False)      # end states only
                assert np.array_equal(end, want)
                # the compact-metadata forms of the same batch (round 4)
                end, bm = dfa.exec_batch_offsets32(base, off.astype(np.uint32))
                assert np.array_equal(end, want) and np.array_equal(bits(bm, len(strings)), ret == 1), (name, cname, L, mode, waves, "off32")
                end, bm = dfa.exec_batch_lengths(base, lens)
                assert np.array_equal(end, want) and np.array_equal(bits(bm, len(strings)), ret == 1), (name, cname, L, mode, waves, "lengths")
                _, bm = dfa.exec_batch_lengths(base, lens, want_end=False)          # the 1-bit-per-input answer alone
                assert np.array_equal(bits(bm, len(strings)), ret == 1)
    for _, dfa in dfas:
        dfa.close()


def test_packed_kernel_device_front_alignments(hip):
    """Device-resident batches of exactly the inputs' size: base at every byte alignment within a line, off[0] != 0,
    the offsets array at a 16-byte and at an 8-mod-16 address, totals that end exactly on a 128-byte line, bitmap
    only (the kernel's own code scratch), every total from 0 to 40 bytes."""
    import torch
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c1.npz"))
    o = Oracle(g.flat)
    rng = np.random.RandomState(19)
    a = np.frombuffer(b"Llibfsmx\0", np.uint8)

    def tiny(k):
        s = bytearray(bytes(a[rng.randint(0, len(a), k)]))
        if k >= 4 and rng.randint(2):
            at = rng.randint(0, k - 3)
            s[at:at + 4] = b"libf"
        return bytes(s)

    dfa = hip.HipDfa(g.flat)
    batches = []
    for total in range(0, 41):
        cut = sorted(rng.randint(0, total + 1, rng.randint(0, 6)))
        body = tiny(total)
        batches.append([body[x:y] for x, y in zip([0] + cut, cut + [total])])
    for k in range(40):
        batches.append([tiny(rng.randint(0, 70)) for _ in range(rng.randint(1, 400))])
    for total in (128, 256, 1024, 4096):                                  # the batch ends exactly on a line (and a row) boundary
        strs = [tiny(rng.randint(1, 30)) for _ in range(total // 16)]
        body = b"".join(strs)[:total - 3]
        batches.append([body[i:i + 7] for i in range(0, len(body), 7)] + [b"lib", b"", b""])
    for bi, strings in enumerate(batches):
        ret, want = o.exec_strings(strings)
        base, off = _packed(strings)
        n = len(strings)
        for shift, lead, osh in ((0, 0, 0), (1, 0, 1), (13, 5, 0), (127, 300, 1), (64, 128, 0)):
            if bi % 3 and (shift, lead) != (0, 0):
                continue
            big = torch.zeros(256 + shift + lead + len(base), dtype=torch.uint8, device="cuda")     # 256-byte aligned allocation
            if len(base):
                big[shift + lead:shift + lead + len(base)] = torch.from_numpy(base.copy()).cuda()
            d_off_store = torch.zeros(n + 3, dtype=torch.int64, device="cuda")
            d_off = d_off_store[osh:osh + n + 1]
            d_off.copy_(torch.from_numpy((off + np.uint64(lead)).view(np.int64)).cuda())
            d_end = torch.full((n,), 7, dtype=torch.int32, device="cuda")
            d_bm = torch.zeros((n + 63) // 64 + 1, dtype=torch.int64, device="cuda")
            base_ptr = big.data_ptr() + shift
            d_off32 = (d_off - lead).to(torch.int32)                       # the u32 form is relative to the first input's byte
            d_len = (d_off[1:] - d_off[:-1]).to(torch.int32)
            for waves, mode in ((0, -1), (1, hip.IN_GENERIC), (0, hip.IN_RAGGED)):
                dfa.tune(hip.KNOB_WAVES, waves)
                dfa.tune(hip.KNOB_INPUT_MODE, mode)
                for form in ("off32", "len"):
                    d_end.fill_(7)
                    d_bm.zero_()
                    if form == "off32":
                        dfa.exec_batch_offsets32_device(base_ptr + lead, d_off32.data_ptr(), n, d_end.data_ptr(), d_bm.data_ptr())
                    else:
                        dfa.exec_batch_lengths_device(base_ptr + lead, d_len.data_ptr(), n, d_end.data_ptr(), d_bm.data_ptr())
                    torch.cuda.synchronize()
                    assert np.array_equal(d_end.cpu().numpy().view(np.uint32), want), (bi, n, len(base), shift, lead, osh, waves, mode, form)
                    assert np.array_equal(bits(d_bm.cpu().numpy()[:(n + 63) // 64], n), ret == 1)
                d_end.fill_(7)
                d_bm.zero_()
                dfa.exec_batch_offsets_device(base_ptr, d_off.data_ptr(), n, d_end.data_ptr(), d_bm.data_ptr())
                torch.cuda.synchronize()
                assert np.array_equal(d_end.cpu().numpy().view(np.uint32), want), (bi, n, len(base), shift, lead, osh, waves)
                assert np.array_equal(bits(d_bm.cpu().numpy()[:(n + 63) // 64], n), ret == 1)
                d_bm.zero_()
                dfa.exec_batch_offsets_device(base_ptr, d_off.data_ptr(), n, 0, d_bm.data_ptr())      # bitmap only
                torch.cuda.synchronize()
                assert np.array_equal(bits(d_bm.cpu().numpy()[:(n + 63) // 64], n), ret == 1), (bi, "bitmap only")
                assert int(d_bm[-1]) == 0
    dfa.close()


def test_packed_kernel_large_batch_and_auto_choice(hip):
    """6e6 packed inputs of 8..64 bytes resident on the device (216 MB): the device-side choice (walk_generic: mean 36
    bytes) agrees with walk_ragged and walk_generic forced on every input and with the oracle on a sample; 2e6 inputs of
    0..1024 bytes (mean 512: the choice is walk_ragged) the same; and the u32-offsets and lengths-only forms agree too."""
    import torch
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c1.npz"))
    o = Oracle(g.flat)
    rng = np.random.RandomState(3)
    dfa = hip.HipDfa(g.flat)
    for n, lo, hi in ((6_000_000, 8, 65), (2_000_000, 0, 1025)):
        lens = rng.randint(lo, hi, n).astype(np.int64)
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum(lens)
        total = int(off[-1])
        buf = torch.empty(((total + 1023) // 1024 + 1, 1024), dtype=torch.uint8, device="cuda")
        hip.gen_inputs_device(buf.data_ptr(), buf.shape[0], 1024, 0, 0x5EEDF5A1, None, b"Libfsm", 2)
        d_off = torch.from_numpy(off.view(np.int64)).cuda()
        e = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(3)]
        bm = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
        ms = {}
        for k, mode in enumerate((-1, hip.IN_RAGGED, hip.IN_GENERIC)):
            dfa.tune(hip.KNOB_INPUT_MODE, mode)
            for rep in range(2):
                dfa.exec_batch_offsets_device(buf.data_ptr(), d_off.data_ptr(), n, e[k].data_ptr(), bm.data_ptr() if k == 0 else 0)
            ms[mode] = dfa.last_kernel_ms()
        torch.cuda.synchronize()
        assert torch.equal(e[0], e[2]) and torch.equal(e[1], e[2])
        dfa.tune(hip.KNOB_INPUT_MODE, -1)
        d_off32, d_len = d_off.to(torch.int32), (d_off[1:] - d_off[:-1]).to(torch.int32)
        bm2 = torch.zeros_like(bm)
        dfa.exec_batch_offsets32_device(buf.data_ptr(), d_off32.data_ptr(), n, e[1].data_ptr(), bm2.data_ptr())
        ms["off32"] = dfa.last_kernel_ms()
        assert torch.equal(e[1], e[2]) and torch.equal(bm2, bm)
        e[1].fill_(5)
        bm2.zero_()
        dfa.exec_batch_lengths_device(buf.data_ptr(), d_len.data_ptr(), n, e[1].data_ptr(), bm2.data_ptr())
        ms["len"] = dfa.last_kernel_ms()
        torch.cuda.synchronize()
        assert torch.equal(e[1], e[2]) and torch.equal(bm2, bm)
        bm2.zero_()
        dfa.exec_batch_lengths_device(buf.data_ptr(), d_len.data_ptr(), n, 0, bm2.data_ptr())      # bitmap only
        ms["len_bitmap"] = dfa.last_kernel_ms()
        assert torch.equal(bm2, bm)
        assert int((e[0] != -1).sum()) == int(bits(bm.cpu().numpy(), n).sum()) > 0
        host = buf.reshape(-1)[:total].cpu().numpy()
        idx = rng.randint(0, n, 5000)
        strings = [bytes(host[int(off[i]):int(off[i + 1])]) for i in idx]
        ret, want = o.exec_strings(strings)
        assert np.array_equal(e[0].cpu().numpy().view(np.uint32)[idx], want)
        print(f"packed front n={n} lens={lo}..{hi - 1}: auto {total / ms[-1] / 1e6:.0f} GB/s, ragged {total / ms[hip.IN_RAGGED] / 1e6:.0f}, generic {total / ms[hip.IN_GENERIC] / 1e6:.0f}, "
              f"u32 offsets {total / ms['off32'] / 1e6:.0f}, lengths only {total / ms['len'] / 1e6:.0f} (walk kernel; bitmap only {total / ms['len_bitmap'] / 1e6:.0f})")
    dfa.close()


# ---------------------------------------------------------------------------
# the C-ABI matrix: ids / resume / eager over packed offsets, host and device pointers
# ---------------------------------------------------------------------------

def test_offsets_fronts_ids_resume_eager(hip):
    """fsm_hip_exec_batch_{ids,resume,eager}_offsets[_device]: the packed form of the three fronts that only had the
    fixed-stride form.  ids against the oracle's fsm_endid_get sets (EARLIEST / RET), resume by cutting every line in two
    pieces (the state after piece 1 carried into piece 2 = the whole line's result), eager on every tests/eager_output
    automaton against the golden id sets -- short lines (per-lane kernel) and long ones (ragged kernel), host and
    device pointers."""
    import torch
    from common import eager_golden_paths
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c3.npz"))
    o = Oracle(g.flat)
    rng = np.random.RandomState(4)
    a = np.frombuffer(b"abcdwxyz0123456789", np.uint8)
    pats = bytes(np.load(os.path.join(GOLDEN, "c3.npz"))["patterns"]).split(b"\n")

    def line(k):
        if rng.randint(3):
            return bytes(a[rng.randint(0, len(a), k)])
        p = pats[rng.randint(len(pats))]
        return p[1:p.index(b"[")] + bytes(rng.randint(48, 58, max(1, k - 6)).astype(np.uint8)) + b"yz"

    dfa = hip.HipDfa(g.flat)
    sets = dfa.ret_sets()
    for lo, hi, n in ((0, 60, 7000), (0, 700, 3000)):
        strings = [line(rng.randint(lo, hi)) for _ in range(n)]
        ret, want = o.exec_strings(strings)
        assert (ret == 1).sum() > n // 10
        base, off = _packed(strings)
        d_base = torch.from_numpy(np.concatenate([base, np.zeros(1, np.uint8)])).cuda()
        d_off = torch.from_numpy(off.view(np.int64)).cuda()
        for mode in (1, 2):
            ids = dfa.exec_offsets_ids(base, off, mode)
            d_ids = torch.full((n,), 5, dtype=torch.int32, device="cuda")
            dfa.exec_offsets_device_front("ids", d_base.data_ptr(), d_off.data_ptr(), n, d_ids.data_ptr(), mode=mode)
            torch.cuda.synchronize()
            assert np.array_equal(d_ids.cpu().numpy().view(np.uint32), ids)
            assert (ids[want == NO] == NO).all()
            for i in np.nonzero(want != NO)[0][:400]:
                e = o.endids(int(want[i]))
                if mode == 1:
                    assert ids[i] == (int(e[0]) if len(e) else 0xFFFFFFFE)
                else:
                    assert np.array_equal(sets[ids[i]], e)
        # resume: piece 1 = the first half of every line, piece 2 = the rest
        cut = [rng.randint(0, len(s) + 1) for s in strings]
        b1, o1 = _packed([s[:c] for s, c in zip(strings, cut)])
        b2, o2 = _packed([s[c:] for s, c in zip(strings, cut)])
        st, _ = dfa.exec_offsets_resume(b1, o1, np.full(n, hip.STATE_START, np.uint32))
        st2, end2 = dfa.exec_offsets_resume(b2, o2, st)
        assert np.array_equal(end2, want)
        d_st = torch.from_numpy(st.view(np.int32)).cuda()
        d_end = torch.full((n,), 5, dtype=torch.int32, device="cuda")
        d_b2 = torch.from_numpy(np.concatenate([b2, np.zeros(1, np.uint8)])).cuda()
        d_o2 = torch.from_numpy(o2.view(np.int64)).cuda()
        dfa.exec_offsets_device_front("resume", d_b2.data_ptr(), d_o2.data_ptr(), n, d_st.data_ptr(), d_end.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d_end.cpu().numpy().view(np.uint32), want) and np.array_equal(d_st.cpu().numpy().view(np.uint32), st2)
    dfa.close()
    # eager outputs: the golden automata, their own inputs packed (+ each input repeated to make long lines)
    for path in eager_golden_paths()[:12]:
        ge = Golden(path)
        strs = ge.strings()
        d = hip.HipDfa(ge.flat)
        base, off = _packed(strs)
        end, sets_ = d.exec_offsets_eager(base, off)
        rows, lens = ge.padded_rows()
        end_r, sets_r = d.exec_batch_eager(rows, lens)
        assert np.array_equal(end, end_r)
        for i in range(len(strs)):
            assert np.array_equal(np.sort(sets_[i]), np.sort(ge.eager_of(i))), (path, i)
            assert np.array_equal(np.sort(sets_[i]), np.sort(sets_r[i]))
        d.close()


# ---------------------------------------------------------------------------
# multi-device front: descriptor form, asynchronous exchange, every visible device
# ---------------------------------------------------------------------------

def _node_device_lists(hip):
    import torch
    out = [[0], [0, 0]]
    if torch.cuda.device_count() > 1:          # switches itself on the moment the box has more than one GPU
        out.append(list(range(torch.cuda.device_count())))
    return out


def test_node_front_descriptor_lengths_offsets_ids_async(hip):
    """fsm_hip_node_exec_device: device-resident shards with lengths, with packed offsets, with device-delivered ids; the
    asynchronous form with two sets of buffers (step k's exchange under step k + 1's walk) and fsm_hip_node_wait; a
    NULL bitmap entry is refused.  On [0] (RCCL, a communicator of one), [0, 0] (peer copies) and -- on a box with
    several GPUs -- on all of them over RCCL."""
    import torch
    import bench
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c3.npz"))
    o = Oracle(g.flat)
    for devices in _node_device_lists(hip):
        node = hip.HipNode(g.flat, devices)
        G = node.ndev
        n, L = 100_037, 256
        host = bench.generate_host(hip, "c3", n, L, 0)
        rng = np.random.RandomState(G)
        lens = np.where(rng.randint(0, 4, n) == 0, rng.randint(0, L + 1, n), L).astype(np.uint32)
        ret_l, want_l = o.exec_stride(host, lens)
        want = o.table_walk(host)
        W = node.bitmap_words(n)
        bufs, dlens, ends, ids, bms, bms2, doffs, packs = [], [], [], [], [], [], [], []
        for k, dv in enumerate(devices):
            f, c = node.shard(n, k)
            dev = f"cuda:{dv}"
            bufs.append(torch.from_numpy(host[f:f + max(c, 1)].copy()).to(dev) if c else torch.zeros((1, L), dtype=torch.uint8, device=dev))
            dlens.append(torch.from_numpy(lens[f:f + max(c, 1)].view(np.int32).copy()).to(dev))
            ends.append(torch.full((max(c, 1),), -2, dtype=torch.int32, device=dev))
            ids.append(torch.full((max(c, 1),), -2, dtype=torch.int32, device=dev))
            bms.append(torch.full((W,), -1, dtype=torch.int64, device=dev))
            bms2.append(torch.full((W,), -1, dtype=torch.int64, device=dev))
            # the shard once more as packed lines: the first lens[i] bytes of every row
            rows_k = [bytes(host[i, :lens[i]]) for i in range(f, f + c)]
            pb, po = _packed(rows_k) if c else (np.zeros(0, np.uint8), np.zeros(1, np.uint64))
            packs.append(torch.from_numpy(np.concatenate([pb, np.zeros(16, np.uint8)])).to(dev))
            doffs.append(torch.from_numpy(po.view(np.int64).copy()).to(dev))
        for dv in set(devices):
            torch.cuda.synchronize(dv)
        P = lambda ts: [t.data_ptr() for t in ts]          # noqa: E731

        def gathered(ts):
            return np.concatenate([ts[k][:node.shard(n, k)[1]].cpu().numpy().view(np.uint32) for k in range(G)])

        # stride + lengths, ids (EARLIEST) and the bitmap + count in one call
        cnt = node.exec_device(n, P(bufs), stride=L, d_len=P(dlens), d_end=P(ends), d_ids=P(ids), ids_mode=1, d_bitmap_all=P(bms), want_count=True)
        assert np.array_equal(gathered(ends), want_l) and cnt == int((ret_l == 1).sum())
        got_ids = gathered(ids)
        for i in np.nonzero(want_l != NO)[0][:300]:
            e = o.endids(int(want_l[i]))
            assert got_ids[i] == (int(e[0]) if len(e) else 0xFFFFFFFE)
        for k in range(G):
            assert np.array_equal(bits(bms[k].cpu().numpy(), n), ret_l == 1), (devices, k)
        # the same inputs as packed lines
        for e_ in ends:
            e_.fill_(-2)
        node.exec_device(n, P(packs), d_off=P(doffs), d_end=P(ends), d_bitmap_all=P(bms2))
        assert np.array_equal(gathered(ends), want_l)
        assert np.array_equal(bits(bms2[0].cpu().numpy(), n), ret_l == 1)
        # asynchronous: three steps alternating two bitmap sets; whole rows
        for step in range(3):
            node.exec_device(n, P(bufs), stride=L, d_end=P(ends), d_bitmap_all=P(bms if step % 2 == 0 else bms2), want_count=True, async_=True)
        cnt = node.wait(want_count=True)
        assert cnt == int((want != NO).sum())
        assert np.array_equal(gathered(ends), want)
        for k in range(G):
            assert np.array_equal(bits(bms[k].cpu().numpy(), n), want != NO) and np.array_equal(bits(bms2[k].cpu().numpy(), n), want != NO)
        # a NULL bitmap entry: refused, nothing launched
        if G > 1:
            with pytest.raises(OSError):
                node.exec_device(n, P(bufs), stride=L, d_bitmap_all=[bms[0].data_ptr()] + [0] * (G - 1))
        # host-pointer ids over the node
        rows = host[:5000]
        hid = node.exec_batch_ids(rows, 1)
        d1 = hip.HipDfa(g.flat)
        assert np.array_equal(hid, d1.exec_batch_ids(rows, 1))
        d1.close()
        node.close()


def test_bench_one_rank_per_visible_gpu():
    """bench.py launched the way the driver launches it for N > 1 -- torch.distributed.run, one rank per GPU, RCCL -- on
    every GPU this box shows.  Skips itself on a one-GPU box (tests/test_gpu_parity.py::test_bench_two_ranks_share_one_gpu
    covers the plumbing there over gloo)."""
    import json
    import subprocess
    import sys
    import torch
    ng = torch.cuda.device_count()
    if ng < 2:
        pytest.skip("one GPU: the RCCL path needs at least two")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ng), "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", str(ng), "--steps", "5", "--warmup", "2", "--workload", "c2",
           "--inputs", "1048576"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == ng and r["value"] > 0 and abs(r["config"]["accepted_inputs"] - ng * 1048576 // 8) < 64 * ng


# ---------------------------------------------------------------------------
# FSM_PRINT_HIP: the reference's rx(1) -> table file -> a libfsm-free matcher
# ---------------------------------------------------------------------------

def _integration_exe(*parts):
    """An integration binary: built where /root/reference exists, prebuilt files elsewhere.  Missing: skip -- or fail
    when FSM_REQUIRE_INTEGRATION is set (a GPU box that is expected to carry the prebuilt files)."""
    exe = os.path.join(ROOT, "integration", "_build", *parts)
    if not os.path.exists(exe):
        if os.environ.get("FSM_REQUIRE_INTEGRATION"):
            pytest.fail("%s missing and FSM_REQUIRE_INTEGRATION is set" % exe)
        pytest.skip("%s not built (needs /root/reference at build time)" % exe)
    return exe


def test_rx_l_hip_into_hipgrep(hip, tmp_path):
    """configs[2] end to end with the reference's own front: rx(1) -- rebuilt with integration/print/print_hip.patch --
    compiles the 1 024 patterns of the C3 workload and prints the DFA with `-l hip`; examples/hipgrep.c (plain C, no
    libfsm in the process) loads the file, matches 20 000 lines in one launch and prints line:end-ids.  The ids must be
    what fsm_exec + fsm_endid_get give on the reference's own union of the same patterns."""
    import subprocess
    from oracle import pyoracle
    rx = _integration_exe("print", "rx")
    if not pyoracle.have_ref():
        pytest.skip("oracle/_ref not built")
    pats = bytes(np.load(os.path.join(GOLDEN, "c3.npz"))["patterns"]).split(b"\n")
    assert len(pats) == 1024
    pf = tmp_path / "patterns"
    pf.write_bytes(b"\n".join(pats) + b"\n")
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([rx, "-u", "-l", "hip", str(pf)], capture_output=True, env=env, timeout=600)
    assert out.returncode == 0 and out.stdout[:6] == b"FSMHIP", out.stderr[-500:]
    table = tmp_path / "c3.fsmhip"
    table.write_bytes(out.stdout)
    rng = np.random.RandomState(8)
    a = np.frombuffer(b"abcdwxyz0123456789", np.uint8)
    lines = []
    for i in range(20000):
        if i % 2:
            p = pats[rng.randint(len(pats))]
            lines.append(p[1:p.index(b"[")] + bytes(rng.randint(48, 58, rng.randint(1, 40)).astype(np.uint8)) + (b"x" if rng.randint(2) else b"yz"))
        else:
            lines.append(bytes(a[rng.randint(0, len(a), rng.randint(0, 50))]))
    exe = str(tmp_path / "hipgrep")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "hipgrep.c"), "-o", exe,
                           "-L" + os.path.join(ROOT, "libfsm_amd"), "-lfsm_hip", "-Wl,-rpath," + os.path.join(ROOT, "libfsm_amd")])
    out = subprocess.run([exe, str(table)], input=b"\n".join(lines) + b"\n", capture_output=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    f = pyoracle.RefFsm.union_res("pcre", pats, 0)
    ret, end = f.exec_strings(lines)
    want = [f"{i + 1}:" + ",".join(str(int(x)) for x in f.endids(int(end[i]))) for i in range(len(lines)) if ret[i] == 1]
    assert out.stdout.decode().split() == want
    assert len(want) >= 9000
