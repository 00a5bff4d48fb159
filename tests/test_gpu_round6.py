"""GPU (-m gpu), round 6: the whole-device front for ONE big input (file.hip), the record walk back on the 32-bit lines
kernel under the harness that once showed it losing a state, and the checks of what the round changed underneath (the
queue-based lazy walk of packed lines, the 2-byte global table)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from common import GOLDEN, Golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


def test_one_big_input_walked_by_the_whole_device(hip, tmp_path):
    """fsm_hip_match_file / fsm_hip_match_buffer_big (file.hip): 1 KiB pieces walked at once from guessed states, the guesses
    corrected until they stand = fsm_exec over the whole input.  Right-anchored patterns (every byte matters, no absorbing
    state): inputs of 300 KB .. 70 MB -- below one window, exactly one, a window and a bit, two and a tail -- ending in a
    pattern or not; the end STATE is compared with the oracle's walk of the same bytes as one input."""
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c3u.npz"))
    o = Oracle(g.flat)
    dfa = hip.HipDfa(g.flat)
    pats = bytes(np.load(os.path.join(GOLDEN, "c3u.npz"))["patterns"]).split(b"\n")
    rng = np.random.RandomState(66)
    alnum = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", np.uint8)
    W = 32 << 20
    for k, size in enumerate((300_000, 1 << 20, W - 1024, W, W + 1, W + 1023, W + 5000, 2 * W + 77_777)):
        data = alnum[rng.randint(0, 36, size)].copy()
        if k % 2 == 0:
            suf = pats[(k * 37) % len(pats)]
            suf = suf[:suf.index(b"[")] + b"4"
            data[size - len(suf):] = np.frombuffer(suf, np.uint8)
        want = o.table_walk(data[None, :])[0]
        r, end = dfa.match_buffer_big(data.tobytes())
        win, passes = dfa.match_last_passes()
        assert end == want and r == int(want != NO), (k, size, end, want)
        assert win >= 1 and passes <= 3 * win, (size, win, passes)        # a pattern automaton forgets: the second pass stands
        if k in (0, 3, 6):
            p = tmp_path / f"big{k}.bin"
            p.write_bytes(data.tobytes())
            assert dfa.match_file(str(p)) == r and dfa.match_buffer(data.tobytes()) == r
    dfa.close()


def test_big_input_on_an_automaton_that_does_not_forget(hip):
    """(aa)*b? over a's: the state after a piece depends on the state before it, every guess but the first is a coin toss and a
    pass puts right one more run of pieces -- the engine still ends where the sequential walk does (and says how many passes)."""
    from oracle.pyoracle import Oracle
    # states: 0 even (end), 1 odd; 'a' flips; anything else: no edge
    nt = np.full((2, 256), -1, np.int64)
    nt[0, ord("a")] = 1
    nt[1, ord("a")] = 0
    flat = hip.FlatDfa.from_dense(nt, 0, [1, 0])
    o = Oracle(flat)
    dfa = hip.HipDfa(flat)
    for size in (300_001, 300_002, (1 << 20) + 3):
        data = np.full(size, ord("a"), np.uint8)
        want = o.table_walk(data[None, :])[0]
        r, end = dfa.match_buffer_big(data.tobytes())
        assert end == want and r == int(size % 2 == 0), (size, end, want)
    data = np.full(400_000, ord("a"), np.uint8)
    data[123_456] = ord("#")                              # a missing edge: DEAD from there on, reading may stop
    r, end = dfa.match_buffer_big(data.tobytes())
    assert r == 0 and end == NO
    dfa.close()


def test_record_walk_on_the_lines_kernel_under_the_harness_that_lost_a_state(hip):
    """Round 5 kept SparsePol off walk_lines32 after one build lost a lane's state in ~45 % of launches
    (profiles/r08i_lines32_sparse_intermittent.txt).  Round 6 took every FLAT load out of the record walk (explicit LDS / global
    address spaces) and put the instantiation back; this is that harness -- the same lines, every workgroup size, with and
    without the accept bitmap -- at 100 launches a row."""
    env = dict(os.environ, REPS="100")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "repro_sparse.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-500:]
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("waves")]
    assert len(rows) == 6 and all("walk_lines32<fsmhip::SparsePol" in ln and " 0/100, end states only 0/100" in ln for ln in rows), r.stdout


def _golden_paths(sub):
    import glob
    return sorted(glob.glob(os.path.join(GOLDEN, sub, "*.npz")))


def test_many_dfa_front_delivers_end_ids(hip):
    """fsm_hip_exec_multi_ids: the many-DFA submission with end-ids by the device -- what re(1) -z and the generated matchers'
    `unsigned *id` hand their callers (src/re/main.c:1152-1166, src/libfsm/print/c.c:569-619).  The reference's endids and
    re_strings goldens + the C3 union (ids = patterns, 238 states with several), one submission: EARLIEST and RET against each
    dfa's own fsm_hip_exec_batch_ids; FSM_HIP_IDS_ERROR refuses the submission when a dfa has an ambiguous end state."""
    gs = [Golden(p) for p in [os.path.join(GOLDEN, n) for n in ("endids_union_det.npz", "endids_union_min.npz", "re_strings_1.npz", "re_strings_2.npz", "c3.npz")]]
    gs += [Golden(p) for p in _golden_paths("retest")[:8]]
    dfas = [hip.HipDfa(g.flat, hip.DEFER_UPLOAD) for g in gs]
    jobs = [g.strings() for g in gs]
    for mode in (1, 2):
        outs = hip.exec_multi_ids(dfas, jobs, mode)
        assert hip.multi_last_launches() == 1 and hip.multi_last_fused_jobs() == len(gs)
        for g, d, strs, (end, bm, ids) in zip(gs, dfas, jobs, outs):
            want_end = np.where(g.ret == 1, g.end, NO).astype(np.uint32)
            assert np.array_equal(end, want_end), g.name
            one = hip.HipDfa(g.flat)                 # the single-dfa front on the same lines
            L = max(1, max((len(x) for x in strs), default=1))
            rows = np.zeros((len(strs), L), np.uint8)
            lens = np.array([len(x) for x in strs], np.uint32)
            for i, x in enumerate(strs):
                rows[i, :len(x)] = np.frombuffer(x, np.uint8)
            assert np.array_equal(ids, one.exec_batch_ids(rows, mode, lens)), (g.name, mode)
            one.close()
    with pytest.raises(OSError):                     # c3 has end states with two ids: AMBIG_ERROR refuses
        hip.exec_multi_ids(dfas, jobs, 3)
    ok = [q for q, g in enumerate(gs) if g.ids_off is None or all(int(g.flat.endid_off[s + 1]) - int(g.flat.endid_off[s]) <= 1 for s in range(g.flat.nstates))]
    assert len(ok) >= 8
    outs = hip.exec_multi_ids([dfas[q] for q in ok], [jobs[q] for q in ok], 3)     # conflict-free: AMBIG_ERROR = EARLIEST
    ref = hip.exec_multi_ids([dfas[q] for q in ok], [jobs[q] for q in ok], 1)
    for (e1, _, i1), (e2, _, i2) in zip(outs, ref):
        assert np.array_equal(e1, e2) and np.array_equal(i1, i2)
    for d in dfas:
        d.close()


def test_many_dfa_device_front_fuses_big_jobs_of_small_automata(hip):
    """64 small automata (the retest goldens, cycled) x 30 000 lines each, device pointers: ONE launch (round 5 fused only jobs of
    up to 65 536 lines and at one wavefront per workgroup); end states and ids against each dfa's own walk of the same lines."""
    import torch
    gs = [Golden(p) for p in _golden_paths("retest")]
    K, n = 64, 30_000
    rng = np.random.RandomState(8)
    dfas, jobs, keep, want = [], [], [], []
    for q in range(K):
        g = gs[q % len(gs)]
        seeds = g.strings() or [b"a"]
        strs = [seeds[rng.randint(len(seeds))] + bytes(rng.randint(32, 127, rng.randint(0, 12)).astype(np.uint8)) for _ in range(n)]
        off = np.zeros(n + 1, np.int64)
        off[1:] = np.cumsum([len(x) for x in strs])
        base = np.frombuffer(b"".join(strs) + b"\0" * 16, np.uint8).copy()
        d = hip.HipDfa(g.flat, hip.DEFER_UPLOAD)
        tb, to = torch.from_numpy(base).cuda(), torch.from_numpy(off).cuda()
        te = torch.full((n,), 7, dtype=torch.int32, device="cuda")
        ti = torch.full((n,), 7, dtype=torch.int32, device="cuda")
        tm = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
        keep += [tb, to, te, ti, tm]
        dfas.append(d)
        jobs.append((tb.data_ptr(), to.data_ptr(), n, te.data_ptr(), tm.data_ptr(), ti.data_ptr()))
        one = hip.HipDfa(g.flat)
        e, _ = one.exec_batch_offsets(base[:off[-1]] if off[-1] else base[:1], off.astype(np.uint64))
        want.append(e)
        one.close()
    hip.exec_multi_ids_device(dfas, jobs, 1)
    torch.cuda.synchronize()
    assert hip.multi_last_launches() == 1 and hip.multi_last_fused_jobs() == K
    for q in range(K):
        te, tm = keep[5 * q + 2], keep[5 * q + 4]
        assert np.array_equal(te.cpu().numpy().view(np.uint32), want[q]), q
        bits = np.unpackbits(tm.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)
        assert np.array_equal(bits, want[q] != NO), q
    for d in dfas:
        d.close()


@pytest.mark.gpu
def test_prepared_many_dfa_submission_replays_from_a_hip_graph(hip):
    """fsm_hip_multi_prepare puts descriptors, tile map and tables on the device once; fsm_hip_multi_launch is one kernel launch
    with no copy, allocation or wait -- captured into a HIP graph and replayed on three different sets of lines in the same
    buffers; end states, bitmap and ids against each dfa's own walk (reperf's repeat loop, src/retest/reperf.c:772-784, for
    all the retest goldens' automata at once)."""
    import torch
    gs = [Golden(p) for p in _golden_paths("retest")]
    K, n, L = len(gs), 700, 24
    rng = np.random.RandomState(12)
    dfas, jobs, keep = [], [], []
    off = (np.arange(n + 1) * L).astype(np.int64)
    for g in gs:
        d = hip.HipDfa(g.flat, hip.DEFER_UPLOAD)
        tb = torch.zeros(n * L + 16, dtype=torch.uint8, device="cuda")
        to = torch.from_numpy(off).cuda()
        te = torch.full((n,), 7, dtype=torch.int32, device="cuda")
        ti = torch.full((n,), 7, dtype=torch.int32, device="cuda")
        tm = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
        keep.append((tb, to, te, ti, tm))
        dfas.append(d)
        jobs.append((tb.data_ptr(), to.data_ptr(), n, te.data_ptr(), tm.data_ptr(), ti.data_ptr()))
    pr = hip.MultiPrepared(dfas, jobs, 1)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        pr.launch(st.cuda_stream)        # warm (nothing to warm: the launch is the kernel alone), then captured
        st.synchronize()
        assert hip.multi_last_launches() == 1 and hip.multi_last_fused_jobs() == K
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            pr.launch(st.cuda_stream)
    for rep in range(3):
        want, wid = [], []
        for q, g in enumerate(gs):
            seeds = g.strings() or [b"a"]
            rows = rng.randint(97, 123, (n, L)).astype(np.uint8)
            for i in range(0, n, 2):        # every other line starts with one of the record's own test strings
                sd = seeds[rng.randint(len(seeds))][:L]
                rows[i, :len(sd)] = np.frombuffer(sd, np.uint8)
            one = hip.HipDfa(g.flat)
            e, _ = one.exec_batch(rows)
            ids = one.exec_batch_ids(rows, 1)
            one.close()
            want.append(e)
            wid.append(ids)
            keep[q][0][:n * L].copy_(torch.from_numpy(rows.reshape(-1)))
            keep[q][2].fill_(7)
            keep[q][3].fill_(7)
            keep[q][4].fill_(-1)
        torch.cuda.synchronize()
        gr.replay()
        torch.cuda.synchronize()
        for q in range(K):
            tb, to, te, ti, tm = keep[q]
            assert np.array_equal(te.cpu().numpy().view(np.uint32), want[q]), (rep, q)
            bits = np.unpackbits(tm.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)
            assert np.array_equal(bits, want[q] != NO), (rep, q)
            assert np.array_equal(ti.cpu().numpy().view(np.uint32), wid[q]), (rep, q)
    pr.close()
    for d in dfas:
        d.close()


@pytest.mark.gpu
def test_bench_one_rank_goes_the_multi_gpu_way_over_rccl():
    """FSM_BENCH_FORCE_DIST=1: bench.py's N > 1 path -- an RCCL communicator, the accept bitmap's all-gather overlapped with the next
    walk, the reductions, the multi_gpu record -- with ONE rank on this box's GPU: the real collective library under the real call
    pattern (two ranks over gloo: tests/test_gpu_parity.py::test_bench_two_ranks_share_one_gpu; RCCL refuses two ranks on one device)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FSM_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--workload", "c2", "--inputs", "1048576"],
                         capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["steps"] == 4 and r["value"] > 0
    assert r["multi_gpu"]["backend"] == "nccl" and r["multi_gpu"]["world_size"] == 1 and r["multi_gpu"].get("rccl_version")
    assert abs(r["config"]["accepted_inputs"] - 1048576 // 8) < 64
    assert "RCCL all-gather" in r["config"]["sharding"]
