"""The eager-output callback STREAM (fsm_hip_exec_batch_eager_trace): order and repeats of fsm_exec's callback calls
(src/libfsm/exec.c:120-144).  CPU: the oracle's restatement against the real reference's raw callback stream;
GPU: the HIP path against the oracle and against the live reference."""
import numpy as np
import pytest

from common import Golden

NO = 0xFFFFFFFF

PATS = [b"apple", b"banana", b"^carrot", b"durian$", b"fig", b"ab+c", b"[0-9]{3}", b"an", b"a"]
WORDS = [b"apple", b"banana", b"carrot", b"durian", b"fig", b"abbbc", b"1234", b"zz", b" ", b"anana", b"aaaa"]


def _need_ref():
    from oracle.pyoracle import have_ref
    if not have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")


def _strings(rng, n):
    return [b" ".join(WORDS[k] for k in rng.randint(0, len(WORDS), rng.randint(0, 9))) for _ in range(n)]


def _same_stream(ids, pos, ref_ids):
    """The reference's order inside ONE state's id set is its table's insertion order (eager_output.c:264-266), ours is
    ascending: equal as sequences once every same-position group is sorted."""
    if len(ids) != len(ref_ids):
        return False
    ref_ids = np.asarray(ref_ids).copy()
    k = 0
    while k < len(ids):
        j = k
        while j < len(ids) and pos[j] == pos[k]:
            j += 1
        if not np.array_equal(np.sort(ref_ids[k:j]), ids[k:j]):
            return False
        k = j
    return True


def test_oracle_trace_equals_the_reference_callback_stream(built):
    """fsm_union_repeated_pattern_group over nine patterns (tests/eager_output/utils.c's construction): literal fsm_exec
    with a callback that records EVERY call, on 600 strings; the oracle's stream is the same sequence, repeats and all."""
    _need_ref()
    from oracle.pyoracle import RefFsm, Oracle
    f = RefFsm.union_repeated("pcre", PATS, 1, False)
    flat = f.flatten()
    rng = np.random.RandomState(5)
    strings = _strings(rng, 600)
    ret, end, cnt, streams = f.exec_eager_stream_strings(strings, cap=256)
    off = np.zeros(len(strings) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in strings])
    base = np.frombuffer(b"".join(strings) + b"\0", np.uint8)
    oret, oend, ocnt, otr = Oracle(flat).exec_eager_trace(base, off=off, cap=256)
    assert np.array_equal(oret, ret) and np.array_equal(oend, end) and np.array_equal(ocnt, cnt)
    assert cnt.max() < 256 and cnt.sum() > 3000
    repeats = 0
    for i in range(len(strings)):
        ids, pos = otr[i]
        assert _same_stream(ids, pos, streams[i]), strings[i]
        repeats += len(ids) - len(set(ids.tolist()))
    assert repeats > 500      # the same id fires again and again: multiplicity is really exercised


def test_oracle_trace_folds_to_the_set_form(built):
    """The stream, de-duplicated in order, is the first-emission list the set-valued oracle entry returns, on every
    tests/eager_output golden program."""
    from common import eager_golden_paths
    from oracle.pyoracle import Oracle
    for path in eager_golden_paths():
        g = Golden(path)
        rows, lens = g.padded_rows()
        o = Oracle(g.flat)
        _, _, sets = o.exec_eager(rows, lens)
        ret, end, cnt, tr = o.exec_eager_trace(rows, lens, cap=512)
        assert np.array_equal(ret, g.ret) and np.array_equal(end, g.end)
        for i in range(len(rows)):
            assert cnt[i] <= 512
            assert np.array_equal(np.unique(tr[i][0]), sets[i])
            assert np.all(np.diff(tr[i][1].astype(np.int64)) >= 0) and (len(tr[i][1]) == 0 or tr[i][1][-1] <= lens[i])


@pytest.fixture(scope="module")
def hip(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    torch.cuda.set_device(0)
    import libfsm_amd
    libfsm_amd.load_library()
    return libfsm_amd


@pytest.mark.gpu
def test_trace_golden_programs_every_front(hip):
    """Every tests/eager_output program + planted random text: ids, positions, counts and end states of the HIP stream
    equal the oracle's, through the stride + lengths, fixed-stride and packed fronts, whatever layout the dfa has;
    a small cap cuts the record list but not the count."""
    from common import eager_golden_paths
    from oracle.pyoracle import Oracle
    rng = np.random.RandomState(3)
    total = 0
    for path in eager_golden_paths():
        g = Golden(path)
        rows, lens = g.padded_rows()
        alpha = np.frombuffer((" ".join(g.meta["patterns"]) + " xyz").encode("latin1"), np.uint8)
        rnd = alpha[rng.randint(0, len(alpha), (300, 48))]
        o = Oracle(g.flat)
        for layout in (hip.LAYOUT_AUTO, hip.LAYOUT_GLOBAL):
            dfa = hip.HipDfa(g.flat, layout)
            for data, ln in ((rows, lens), (rnd, None)):
                ret, end, cnt, tr = o.exec_eager_trace(data, ln, cap=128)
                gend, gcnt, gtr = dfa.exec_batch_eager_trace(data, ln, cap=128)
                assert np.array_equal(gend, end) and np.array_equal(gcnt, cnt), g.meta["source"]
                for i in range(len(data)):
                    assert np.array_equal(gtr[i][0], tr[i][0]) and np.array_equal(gtr[i][1], tr[i][1]), (g.meta["source"], i)
                total += int(cnt.sum())
                # packed front, and a cap of 2
                l2 = ln if ln is not None else np.full(len(data), data.shape[1], np.uint32)
                off = np.zeros(len(data) + 1, np.uint64)
                off[1:] = np.cumsum(l2)
                flatb = np.concatenate([data[i, :l2[i]] for i in range(len(data))] + [np.zeros(1, np.uint8)])
                pend, pcnt, ptr = dfa.exec_batch_eager_trace(flatb, off=off, cap=2)
                assert np.array_equal(pend, end) and np.array_equal(pcnt, cnt)
                for i in range(len(data)):
                    assert np.array_equal(ptr[i][0], tr[i][0][:2]) and np.array_equal(ptr[i][1], tr[i][1][:2])
            dfa.close()
    assert total > 2000


@pytest.mark.gpu
def test_trace_live_reference_and_wide_sets(hip):
    """The live reference's raw callback stream (800 strings, nine patterns) against the HIP stream; then a random DFA
    with 200 distinct ids (the wide-set plan) against the oracle, with an absorbing state that fires on every byte."""
    from oracle.pyoracle import Oracle, have_ref
    from test_gpu_parity import random_eager_dfa
    rng = np.random.RandomState(9)
    if have_ref():
        from oracle.pyoracle import RefFsm
        f = RefFsm.union_repeated("pcre", PATS, 1, False)
        dfa = hip.HipDfa.compile_fsm(f.ptr)
        strings = _strings(rng, 800)
        ret, end, cnt, streams = f.exec_eager_stream_strings(strings, cap=256)
        off = np.zeros(len(strings) + 1, np.uint64)
        off[1:] = np.cumsum([len(s) for s in strings])
        base = np.frombuffer(b"".join(strings) + b"\0", np.uint8)
        gend, gcnt, gtr = dfa.exec_batch_eager_trace(base, off=off, cap=256)
        assert np.array_equal(gend, end) and np.array_equal(gcnt, cnt)
        for i in range(len(strings)):
            assert _same_stream(gtr[i][0], gtr[i][1], streams[i]), strings[i]
        dfa.close()
    flat = random_eager_dfa(rng, 300, 200)
    alpha = np.frombuffer(b"abcdefgh", np.uint8)
    rows = alpha[rng.randint(0, 8, (2000, 96))]
    lens = rng.randint(0, 97, 2000).astype(np.uint32)
    o = Oracle(flat)
    ret, end, cnt, tr = o.exec_eager_trace(rows, lens, cap=64)
    assert cnt.max() > 64                       # some streams are cut: the count still tells
    dfa = hip.HipDfa(flat)
    gend, gcnt, gtr = dfa.exec_batch_eager_trace(rows, lens, cap=64)
    assert np.array_equal(gend, end) and np.array_equal(gcnt, cnt)
    for i in range(len(rows)):
        assert np.array_equal(gtr[i][0], tr[i][0]) and np.array_equal(gtr[i][1], tr[i][1]), i
    # device-pointer front, no positions, no end states
    import torch
    d_rows = torch.from_numpy(rows).cuda()
    d_len = torch.from_numpy(lens.view(np.int32)).cuda()
    d_cnt = torch.zeros(len(rows), dtype=torch.int32, device="cuda")
    d_ids = torch.zeros((len(rows), 8), dtype=torch.int32, device="cuda")
    dfa.exec_batch_eager_trace_device(d_rows.data_ptr(), rows.shape[1], len(rows), 8, d_cnt.data_ptr(), d_ids.data_ptr(), d_len=d_len.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_cnt.cpu().numpy().view(np.uint32), cnt)
    got = d_ids.cpu().numpy().view(np.uint32)
    for i in range(len(rows)):
        k = min(int(cnt[i]), 8)
        assert np.array_equal(got[i, :k], tr[i][0][:k])
    dfa.close()
