#!/usr/bin/env python3
"""bench.py -- the hot path (batched DFA walk) on N GPUs of one node.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c5] [--n INPUTS_PER_GPU] [--subs auto|none]
                    [--scaling weak|strong] [--full-parity]

A "step" is one pass of the walk kernel over one batch of synthetic inputs that is already resident in
HBM (generated on the device, so nothing crosses PCIe), followed -- for N > 1 -- by the RCCL all-gather of
the accept bitmap (asynchronous: it overlaps the next step's kernel; all K gathers finish inside the timed
region).  Inputs are sharded by contiguous global index range, one shard per rank.  Rank 0 prints ONE JSON
line.

Workloads (BASELINE.json configs):
  c3  configs[2], the north star's target config: 1 024 anchored PCRE unioned into one ~4 096-state DFA,
      1e8 x 1 KiB inputs per GPU, half derived from a pattern (prefix + digits + suffix), half random. (default)
  c2  configs[1]: PCRE [Ll]ibf+(sm)* DFA (5 states), 1e8 x 1 KiB random inputs, "Libfsm" planted in every 8th.
  c5  configs[4]: Aho-Corasick DFA of 1e5 literals (8-16 characters over 64 symbols, ~1e6 states,
      right-anchored, end-id = literal), built by the library's own fsm_hip_strings_* builder; 1e7 x 1 KiB
      inputs over the same alphabet, every 8th ending with a literal.  Table > LDS: bound by L2 gather
      requests, not by HBM.
  c3t the transition-dense twin of c3 (not a BASELINE config): 1 024 patterns ^<pfx>([0-9][a-f])+(x|yz)$, half the
      rows alternating digit / letter, so a live row changes state on every byte and no chunk can be skipped.
At N = 1 the line carries the other configs as `sub_results` (each with its own roofline, cpu_baseline and
parity): c3 with the chunk skip disabled (every byte pays its lookup test), c3t, c2, c5.  N > 1 defaults to configs[3]'s shard, 1.25e8 inputs per GPU.
The c2/c3 DFA tables come from tests/golden/{c1,c3}.npz (flattened from the real reference by
tests/golden/make_golden.py); /root/reference is not needed at run time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
_T0 = time.time()

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
SEED = 0x5EEDF5A1
ALNUM = b"abcdefghijklmnopqrstuvwxyz0123456789"
KNOB_NOSKIP = 13


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c3t", "c3u", "lds2", "c5", "c2_short", "c3_short", "c2_ragged", "c3_ragged", "c5_short", "c5_ragged", "c3_eager40"],
                    help="c*_short / c*_ragged: the packed-lines front alone (what the default run reports as sub_results), for profiling")
    ap.add_argument("--subs", default="auto", choices=["auto", "none"],
                    help="auto: at N = 1 also measure the other configs and report them as sub_results")
    ap.add_argument("--n", "--inputs", dest="n", type=int, default=0,
                    help="inputs per GPU (weak) / in all (strong); default 1e8 (N > 1: configs[3]'s 1.25e8 per GPU; c5: 1e7)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--c5-words", type=int, default=100_000, help="c5: number of literals")
    ap.add_argument("--len", type=int, default=1024, help="bytes per input")
    ap.add_argument("--input-mode", type=int, default=-1)
    ap.add_argument("--nb", type=int, default=0)
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--layout", type=int, default=0)
    ap.add_argument("--knob", action="append", default=[], help="K=V: raw fsm_hip_dfa_tune knob (see include/fsm_hip_plan.h)")
    ap.add_argument("--no-early-retire", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="inputs for the CPU baseline (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--node-front", action="store_true",
                    help="measure the C multi-device front (fsm_hip_node_*: one replica + host thread per visible GPU, one process) "
                         "on the main workload and print its JSON; bench.py runs this by itself, in a subprocess, when N = 1 sees several GPUs")
    ap.add_argument("--full-parity", action="store_true",
                    help="(the default for the main workload at N = 1; kept for old command lines)")
    ap.add_argument("--no-full-parity", action="store_true",
                    help="skip the full-N check: after timing, ALL inputs of the main workload are streamed back and every end state is "
                         "compared with the threaded CPU table walker (about 25 s at 1e8 x 1 KiB)")
    return ap.parse_args()


def c3_affixes(which="c3"):
    pats = bytes(np.load(os.path.join(ROOT, "tests", "golden", which + ".npz"))["patterns"]).split(b"\n")
    if which == "c3u":   # unanchored patterns <letters>[0-9]$: no prefix; a row accepted by pattern i ends with its letters + the digit i % 10
        return [b""], [p[:p.index(b"[")] + str(i % 10).encode() for i, p in enumerate(pats)]
    return [p[1:p.index(b"[" if which == "c3" else b"(")] for p in pats], [b"x", b"yz"]


ALPHA64 = b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_-"
_C5 = {}


def c5_words(nwords):
    """configs[4]: seeded literals of 8-16 characters over a 64-symbol alphabet."""
    if nwords not in _C5:
        rng = np.random.RandomState(SEED & 0x7FFFFFFF)
        alpha = np.frombuffer(ALPHA64, np.uint8)
        _C5[nwords] = [bytes(alpha[rng.randint(0, 64, rng.randint(8, 17))]) for _ in range(nwords)]
    return _C5[nwords]


_C5_FLAT = {}
LDS2_ALPHA = b"abcdefghijkl"


def lds2_flat(hip):
    """the pair-table regime (FSM_HIP_LAYOUT_LDS2: one LDS lookup per TWO input bytes): 70 literals of 3-7 letters over a
    12-letter alphabet, unanchored with end-ids (no absorbing state) -- the first case of tests/tools/lds2_probe.py"""
    if "lds2" not in _C5_FLAT:
        rng = np.random.RandomState(len(LDS2_ALPHA) * 131 + 70)
        al = np.frombuffer(LDS2_ALPHA, np.uint8)
        words = sorted(set(bytes(al[rng.randint(0, len(al), rng.randint(3, 8))]) for _ in range(70)))
        _C5_FLAT["lds2"] = hip.FlatDfa.from_strings(words, 0, list(range(len(words))))
    return _C5_FLAT["lds2"]


def c5_flat(hip, nwords):
    """the literal-set automaton of configs[4], built once per run (the oracle keeps one 1 GB dense table per automaton object)"""
    if nwords not in _C5_FLAT:
        words = c5_words(nwords)
        # right-anchored, end-id = literal index: no absorbing accept state, every byte is walked
        _C5_FLAT[nwords] = hip.FlatDfa.from_strings(words, 2, list(range(len(words))))
    return _C5_FLAT[nwords]


def c5_tail_table(words):
    lens = np.array([len(w) for w in words], np.int64)
    W = np.zeros((len(words), 16), np.uint8)
    for i, w in enumerate(words):
        W[i, :len(w)] = np.frombuffer(w, np.uint8)
    return W, lens


def c5_plant_tails(rows, first, words, xp):
    """Every 8th input (by global index) ends with the literal (index * 2654435761) % nwords, so that it is
    accepted by the right-anchored automaton; `rows` is a torch (device) or numpy [n][L] array."""
    W, lens = c5_tail_table(words)
    n, L = rows.shape
    if xp is np:
        idx = np.arange((-first) % 8, n, 8)
        Wd, ld = W, lens
    else:
        idx = xp.arange((-first) % 8, n, 8, device=rows.device)
        Wd, ld = xp.from_numpy(W).to(rows.device), xp.from_numpy(lens).to(rows.device)
    widx = ((idx + first) * 2654435761) % len(words)
    for l in sorted(set(lens.tolist())):
        m = ld[widx] == l
        if bool(m.any()):
            rows[idx[m], L - l:] = Wd[widx[m], :l]


def generate(hip, workload, d_ptr, n, L, first, words=None, buf=None):
    if workload == "c5":
        import torch
        hip.gen_inputs_device(d_ptr, n, L, first, SEED, ALPHA64)
        torch.cuda.synchronize()
        c5_plant_tails(buf, first, words, torch)
        return
    if workload == "c2":
        hip.gen_inputs_device(d_ptr, n, L, first, SEED, None, b"Libfsm", 8)
    elif workload == "lds2":
        hip.gen_inputs_device(d_ptr, n, L, first, SEED, LDS2_ALPHA)
    elif workload == "c3t":   # pattern rows alternate digit / [a-f] after the prefix: a state change on every byte
        pf, sf = c3_affixes("c3t")
        hip.gen_affix_inputs_device(d_ptr, n, L, first, SEED, ALNUM, b"0123456789", pf, sf, 2, body2=b"abcdef")
    elif workload == "c3u":   # every row random over [a-z0-9]; every 2nd one ends with a pattern's letters + a digit
        pf, sf = c3_affixes("c3u")
        al = os.environ.get("FSM_BENCH_C3U_ALPHA", "").encode() or ALNUM       # (a measurement aid: e.g. digits only = a walk that never leaves the first rows)
        hip.gen_affix_inputs_device(d_ptr, n, L, first, SEED, al, al, pf, sf, 2)
    else:
        pf, sf = c3_affixes()
        hip.gen_affix_inputs_device(d_ptr, n, L, first, SEED, ALNUM, b"0123456789", pf, sf, 2)


def generate_host(hip, workload, n, L, first, words=None):
    if workload == "c5":
        rows = hip.gen_inputs_host(n, L, first, SEED, ALPHA64)
        c5_plant_tails(rows, first, words, np)
        return rows
    if workload == "c2":
        return hip.gen_inputs_host(n, L, first, SEED, None, b"Libfsm", 8)
    if workload == "lds2":
        return hip.gen_inputs_host(n, L, first, SEED, LDS2_ALPHA)
    if workload == "c3t":
        pf, sf = c3_affixes("c3t")
        return hip.gen_affix_inputs_host(n, L, first, SEED, ALNUM, b"0123456789", pf, sf, 2, body2=b"abcdef")
    if workload == "c3u":
        pf, sf = c3_affixes("c3u")
        al = os.environ.get("FSM_BENCH_C3U_ALPHA", "").encode() or ALNUM
        return hip.gen_affix_inputs_host(n, L, first, SEED, al, al, pf, sf, 2)
    pf, sf = c3_affixes()
    return hip.gen_affix_inputs_host(n, L, first, SEED, ALNUM, b"0123456789", pf, sf, 2)


def rccl_identity(torch):
    """what the collective library says about itself: its version as torch reports it and the shared object mapped into this
    process (the first N > 1 run has to explain itself from one line)"""
    out = {}
    try:
        v = torch.cuda.nccl.version()
        out["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as e:  # noqa: BLE001
        out["rccl_version"] = "unknown: " + repr(e)[:60]
    try:
        libs = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln or "libnccl" in ln})
        out["rccl_library"] = libs[0] if libs else None
    except OSError:
        out["rccl_library"] = None
    return out


def host_cores():
    """(threads to use, cgroup cpu quota or None): a container may be throttled far below the visible cores."""
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        pass
    if quota:
        ncores = max(1, min(ncores, int(quota + 0.5)))
    return ncores, quota


def sample_indices(n, sample):
    """A seeded STRATIFIED sample over the whole index range [0, n) -- not a prefix, and not a constant stride
    either (the generators are periodic in the index: every 2nd / 8th row is special): one uniformly random row
    out of each of `sample` equal strata, plus the first and last 64 rows (the last tile may be partial) and the
    rows either side of byte offset 2^32."""
    sample = max(1, min(sample, n))
    rng = np.random.RandomState(SEED & 0x7FFFFFFF)
    edges = np.linspace(0, n, sample + 1).astype(np.int64)
    width = np.maximum(1, edges[1:] - edges[:-1])
    idx = [edges[:-1] + (rng.randint(0, 1 << 62, sample, dtype=np.int64) % width), np.arange(min(64, n)), np.arange(max(0, n - 64), n)]
    if n > (1 << 22) + 64:
        idx.append(np.arange((1 << 22) - 32, (1 << 22) + 32))
    idx = np.unique(np.concatenate(idx))
    return idx[idx < n]


def kernels_sha16():
    """A recorded counter file speaks for the kernels it was taken from: sha256 of the device sources, first 16 hex digits
    (tools/rocpd_summary.py stores it; a kernel edit makes bench.py drop `traffic` until tools/profile.sh is rerun)."""
    import hashlib
    h = hashlib.sha256()
    # everything that decides what a launch fetches: the kernels, which kernel / grid / workgroup a launch gets
    # (fsm_hip.hip pick_cfg / waves_by_occupancy, launch.h, kern_*.hip), the table layouts (plan.cpp) and the other fronts
    for f in ("walk_kernels.h", "walk_lazy.h", "walk_aux.h", "launch.h", "fsm_hip.hip", "plan.cpp", "plan.h", "multi.hip",
              "kern_tiny.hip", "kern_lds.hip", "kern_comb.hip", "kern_glob.hip", "kern_glob16.hip"):
        with open(os.path.join(ROOT, "libfsm_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def c5_reference_leg(hip):
    """A reference-timed CPU figure beside C5.  At the bench's size (1e5 literals of 8-16 symbols: ~1e6 states) the
    reference cannot be run in a benchmark's time (cpu_baseline.why_not_reference), so its own matchers are timed on the
    largest literal set it builds in seconds: 1e5 literals of 4-8 letters, ~3e5 states (the automaton of
    tests/test_gpu_parity.py::test_config5_aho_corasick_100k_literals): fsm_exec with the per-call isdfa sweep hoisted
    (derived from exec.c, NOT the reference: the literal fsm_exec is ~3 s per call here) and VM v2, 1 thread; the HIP
    path walks the same rows on the same automaton and must agree on every end state."""
    import threading
    from oracle import pyoracle
    if not pyoracle.have_ref():
        return {"error": "oracle/_ref not built"}
    rng = np.random.RandomState(5)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", np.uint8)
    words = sorted(set(bytes(alpha[rng.randint(0, 26, rng.randint(4, 9))]) for _ in range(100000)))
    box = {}

    def work():                       # ac.c recurses per trie node: needs a big stack
        t0 = time.perf_counter()
        box["f"] = pyoracle.RefFsm.re_strings(words, 0, True)
        box["build_s"] = time.perf_counter() - t0
        box["flat"] = box["f"].flatten()

    threading.stack_size(1 << 30)
    th = threading.Thread(target=work)
    th.start()
    th.join()
    threading.stack_size(0)
    f, flat = box["f"], box["flat"]
    nrows, L = 10000, 1024
    rows = alpha[rng.randint(0, 26, (nrows, L))]
    for i in range(0, nrows, 2):      # end half of the rows on a word so they accept
        w = words[rng.randint(len(words))]
        rows[i, L - len(w):] = np.frombuffer(w, np.uint8)
    hret, hend = f.exec_hoisted_stride(rows)
    t_ho = f.last_seconds
    dfa = hip.HipDfa(flat)
    got, _ = dfa.exec_batch(rows)
    layout = dfa.info()["layout_name"]
    dfa.close()
    out = {"literals": len(words), "dfa_states": flat.nstates, "re_strings_seconds": round(box["build_s"], 1), "inputs": nrows, "input_len": L,
           "fsm_exec_hoisted_value": round(nrows * L / 1e9 / t_ho, 5), "unit": "GB/s", "cores": 1,
           "fsm_exec_hoisted_sample": "fsm_exec with the per-call fsm_all(fsm_isdfa) sweep of exec.c:106-109 removed (derived, NOT the reference), 1 thread",
           "hip_layout": layout, "hip_vs_fsm_exec_hoisted": "bit-exact" if np.array_equal(got, hend) and int((hret == 1).sum()) >= nrows // 2 else "MISMATCH"}
    if os.environ.get("FSM_BENCH_C5_VM", "1") != "0":
        for ver in (1, 2):
            t0 = time.perf_counter()
            vm = f.vm_match_stride(rows, ver)
            out[f"vm_v{ver}_value"] = round(nrows * L / 1e9 / f.last_seconds, 5)
            out[f"vm_v{ver}_compile_seconds"] = round(time.perf_counter() - t0 - f.last_seconds, 1)
            out[f"vm_v{ver}_vs_fsm_exec"] = "agree" if np.array_equal(vm == 1, hret == 1) else "MISMATCH"
        out["vm_sample"] = "reference fsm_vm_compile + fsm_vm_match_buffer, 1 thread, the same inputs"
        if out["vm_v2_vs_fsm_exec"] != "agree":
            # found by this leg in round 3: the reference's v2 encoder keeps the index into its far-branch address table in the
            # instruction's 16-bit dest field (src/libfsm/vm/v2.c:71, :129-131: `uint16_t dest_arg ... dest_arg = alen++`), so a
            # program with more than 65 535 far branches jumps to the wrong states; v1 has no such table and agrees with
            # fsm_exec.  From ~5 000 states up (re_strings automata) v2 is not a baseline: its rate is listed, not its answers
            out["vm_v2_note"] = ("the reference's v2 encoding truncates far-branch table indices to 16 bits (vm/v2.c:129-131): "
                                 "wrong answers beyond 65 535 far branches; v1 agrees with fsm_exec and is the VM baseline here")
    return out


def cpu_baseline(hip, workload, flat, L, rows, gpu_end_sample, words=None):
    """Time the reference's CPU paths on the sampled rows (the very bytes the GPU walked, copied back from the
    device; rank 0, N = 1 only) and check the GPU's answers on them against it, bit for bit.
    SURVEY.md 8(d)'s four CPU lines: (1) literal fsm_exec, (2) fsm_exec with the per-call isdfa sweep hoisted
    (derived, non-reference), (3) VM v2 on 1 thread and on every granted core, (4) the gcc-compiled matcher
    `retest -l vmc` generates."""
    from oracle import pyoracle
    out = {"cores": 1, "unit": "GB/s"}
    gb = rows.size / 1e9
    nrows = len(rows)
    if workload in ("c5", "lds2") or not pyoracle.have_ref():   # (lds2: a literal set the library's own builder made -- the automaton is given, not derived from a regex the reference could compile)
        o = get_oracle(flat)
        want = o.table_walk(rows)
        out.update(kind="port", value=round(gb / o.last_seconds, 5),
                   sample=f"oracle dense-table walker (oracle/dfa_oracle.c), 1 thread, {nrows} inputs x {L} B spread over the whole batch")
        if workload == "c5":
            out["why_not_reference"] = ("at this size the reference's fsm_exec takes 3.7 s per 1 KiB input (fsm_all(fsm_isdfa) over ~1e6 states on "
                                        "every call, exec.c:106), re_strings 59 s + 7.5 GB to build the DFA and fsm_vm_compile 7 min + 15 GB "
                                        "(measured in the build container, DESIGN.md section 3); the reference is compared at 3e5 states in "
                                        "tests/test_gpu_parity.py::test_config5_aho_corasick_100k_literals and timed below")
            try:
                out["reference_at_3e5_states"] = c5_reference_leg(hip)
            except Exception as e:  # noqa: BLE001
                out["reference_at_3e5_states"] = {"error": repr(e)[:300]}
        ok = bool(np.array_equal(gpu_end_sample, want))
        if workload == "c5" and out["reference_at_3e5_states"].get("hip_vs_fsm_exec_hoisted") == "MISMATCH":
            ok = False
        return out, ok
    # the real reference, rebuilt as a struct fsm from its own regex sources
    if workload == "c2":
        f = pyoracle.RefFsm.re_comp("pcre", b"[Ll]ibf+(sm)*", 0, True, True, endid=0)
        nfe, nho = nrows, nrows
    else:
        pats = bytes(np.load(os.path.join(ROOT, "tests", "golden", workload + ".npz"))["patterns"]).split(b"\n")
        f = pyoracle.RefFsm.union_res("pcre", pats, 0)
        nfe, nho = min(nrows, 1500), min(nrows, 40000)  # fsm_exec re-runs fsm_all(isdfa) per call: ~9 ms/call on 4k states
    fe_idx = np.linspace(0, nrows - 1, nfe).astype(np.int64)      # the fsm_exec subset is spread over the sample too
    ret, end = f.exec_stride(rows[fe_idx])
    t_exec = f.last_seconds
    ho_idx = np.linspace(0, nrows - 1, nho).astype(np.int64)
    hret, hend = f.exec_hoisted_stride(rows[ho_idx])
    t_hoist = f.last_seconds
    vm = f.vm_match_stride(rows, 2)
    t_vm = f.last_seconds
    want = get_oracle(flat).table_walk(rows)
    assert np.array_equal(end, want[fe_idx]), "oracle != reference fsm_exec"
    assert np.array_equal(hend, want[ho_idx]), "hoisted fsm_exec != fsm_exec"
    ver = 2
    if not np.array_equal(vm == 1, want != 0xFFFFFFFF):
        # the reference's v2 encoder keeps the index into its far-branch address table in the instruction's 16-bit dest field
        # (src/libfsm/vm/v2.c:71, :129-131): an automaton with more than 65 535 far branches -- the complete DFA of an unanchored
        # pattern list (c3u) is one -- is mis-encoded.  v1 has no such field: it is the reference's VM for this automaton.
        vm = f.vm_match_stride(rows, 1)
        t_vm = f.last_seconds
        assert np.array_equal(vm == 1, want != 0xFFFFFFFF), "reference VM (v1 and v2) != fsm_exec"
        ver = 1
        out["vm_v2_vs_fsm_exec"] = "MISMATCH (src/libfsm/vm/v2.c:129-131: 16-bit far-branch index); the VM lines below are v1's"
    # all host cores: the reference itself is single-threaded, so its fastest in-process matcher (VM v2) is run
    # on one thread per core over slices of the sample, repeated to ~1-2 s of wall time
    ncores, quota = host_cores()
    f.match_threads(rows, ncores, 1, ver)                       # untimed pass (burst credit, page faults)
    probe, _ = f.match_threads(rows, ncores, 2, ver)            # sizes the timed run (~2 s)
    reps = max(1, min(2000, int(2.0 * probe / gb)))
    allc, acc = f.match_threads(rows, ncores, reps, ver)
    assert acc == int((want != 0xFFFFFFFF).sum()), "threaded VM run disagrees"
    out.update(kind="reference", value=round(nfe * L / 1e9 / t_exec, 5),
               sample=f"reference fsm_exec (src/libfsm/exec.c) with a (ptr,len) getc, 1 thread, {nfe} inputs x {L} B spread over the whole batch",
               fsm_exec_hoisted_value=round(nho * L / 1e9 / t_hoist, 5),
               fsm_exec_hoisted_sample=f"fsm_exec with the per-call fsm_all(fsm_isdfa) sweep of exec.c:106-109 removed (derived from exec.c by oracle/build_ref.sh: NOT the reference), 1 thread, {nho} inputs",
               vm_v2_value=round(gb / t_vm, 5), vm_v2_sample=f"reference fsm_vm_match_buffer v{ver}, 1 thread, {nrows} inputs", vm_version=ver,
               vm_v2_allcores_value=round(allc, 3), vm_v2_allcores_cores=ncores, vm_v2_allcores_cgroup_cpu_quota=quota,
               vm_v2_allcores_sample=f"same VM shared read-only by {ncores} threads, each walking its slice of the {nrows}-input sample {reps}x")
    # (4) what `retest -l vmc` runs: fsm_print(FSM_PRINT_VMC) -> cc -> dlopen -> fsm_main(b, e) per input
    try:
        m = f.codegen_match_stride(rows, "vmc", ("-O3",) if flat.nstates < 1000 else ("-O1",), timeout=90.0)
        if m is not None:
            assert np.array_equal(m == 1, want != 0xFFFFFFFF), "generated matcher != fsm_exec"
            out.update(codegen_vmc_value=round(gb / f.last_seconds, 5),
                       codegen_vmc_sample=f"the matcher `retest -l vmc` builds (fsm_print FSM_PRINT_VMC, io = pair, gcc {'-O3' if flat.nstates < 1000 else '-O1'}: "
                                          f"{f.last_compile_seconds:.1f} s to print + compile), 1 thread, {nrows} inputs")
    except AssertionError:
        raise
    except Exception as e:  # no C compiler on the box, ...: the optional line is left out
        out["codegen_vmc_error"] = repr(e)[:200]
    parity = bool(np.array_equal(gpu_end_sample, want))
    return out, parity


_ORACLES = {}


def get_oracle(flat):
    """one checker (and one dense table: 1 GB for the 1e5-literal automaton) per automaton and run"""
    from oracle import pyoracle
    if id(flat) not in _ORACLES:
        _ORACLES[id(flat)] = pyoracle.Oracle(flat)
    return _ORACLES[id(flat)]


def full_parity(torch, flat, buf, end, n, L):
    """'full N compare GPU vs CPU table-walker'.  Every input is streamed back from the device
    in 2 GiB slices and walked by the oracle's dense-table walker (itself pinned to fsm_exec on every golden
    fixture) on all granted host cores; every end state must agree."""
    o = get_oracle(flat)
    ncores, _ = host_cores()
    step = max(1, (2 << 30) // L)
    bad, t0, t_cpu = 0, time.perf_counter(), 0.0
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        rows = buf[lo:hi].cpu().numpy()
        want = o.table_walk_mt(rows, ncores)
        t_cpu += o.last_seconds
        got = end[lo:hi].cpu().numpy().view(np.uint32)
        bad += int((got != want).sum())
    return {"rows": n, "mismatches": bad, "cpu_threads": ncores, "seconds": round(time.perf_counter() - t0, 1),
            "cpu_walk_GBps": round(n * L / 1e9 / t_cpu, 2), "checker": "oracle dense-table walker (oracle/dfa_oracle.c), all inputs"}


def node_front(a):
    """The C host's way to use a whole node (include/fsm_hip.h fsm_hip_node_*): ONE process, one table replica and one
    host thread per visible GPU, device-resident shards by contiguous global index range, ONE in-place ncclAllGather of
    the accept bitmap + one ncclAllReduce of the match count per step (RCCL bound by dlopen inside libfsm_hip.so).
    Wall-clock around K calls of fsm_hip_node_exec_batch_device, which returns after every device has finished."""
    import torch
    import libfsm_amd as hip
    hip.load_library()
    L = a.len
    wl = a.workload if a.workload in ("c2", "c3", "c3t", "c3u") else "c3"
    flat = hip.FlatDfa.load(os.path.join(ROOT, "tests", "golden", "c1.npz" if wl == "c2" else wl + ".npz"))
    ndev = torch.cuda.device_count()
    devices = list(range(ndev))
    if os.environ.get("FSM_BENCH_NODE_REPLICAS"):      # a rig with fewer GPUs than replicas: repeat device 0
        devices = [0] * int(os.environ["FSM_BENCH_NODE_REPLICAS"])
    node = hip.HipNode(flat, devices)
    per = a.n if a.n > 0 else 100_000_000
    n = per * len(devices) // (64 * len(devices)) * (64 * len(devices))
    W = node.bitmap_words(n)
    bufs, ends, bms = [], [], []
    for k, dv in enumerate(devices):
        f, c = node.shard(n, k)
        torch.cuda.set_device(dv)
        b = torch.empty((max(c, 1), L), dtype=torch.uint8, device=f"cuda:{dv}")
        generate(hip, wl, b.data_ptr(), c, L, f)
        bufs.append(b)
        ends.append(torch.empty(max(c, 1), dtype=torch.int32, device=f"cuda:{dv}"))
        bms.append(torch.zeros(W, dtype=torch.int64, device=f"cuda:{dv}"))
    for dv in set(devices):
        torch.cuda.synchronize(dv)
    torch.cuda.set_device(devices[0])
    args = ([b.data_ptr() for b in bufs], L, n, [e.data_ptr() for e in ends], [m.data_ptr() for m in bms])
    for _ in range(4 + a.warmup):
        cnt = node.exec_batch_device(*args, want_count=True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        cnt = node.exec_batch_device(*args, want_count=True)
    el = time.perf_counter() - t0
    per_dev_ms = [round(node.replica(k).last_kernel_ms(), 4) for k in range(len(devices))]
    # the asynchronous form: two sets of bitmaps, step k's exchange under step k + 1's walk, one wait at the end
    bms2 = [torch.zeros_like(m) for m in bms]
    t0 = time.perf_counter()
    for k in range(a.steps):
        node.exec_device(n, args[0], stride=L, d_end=args[3], d_bitmap_all=[m.data_ptr() for m in (bms if k % 2 == 0 else bms2)], want_count=True, async_=True)
    cnt_async = node.wait(want_count=True)
    el_async = time.perf_counter() - t0
    # every replica holds the whole bitmap; its popcount is the reduced match count
    same = all(int(np.unpackbits(m.cpu().numpy().view(np.uint8)).sum()) == cnt for m in bms[:2]) and cnt_async == cnt
    out = {"front": "fsm_hip_node_exec_batch_device (C ABI, one process, one host thread per device)", "devices": devices,
           "uses_rccl": node.uses_rccl(), "rccl_library": node.rccl_path(), "workload": wl, "inputs_total": n, "input_len": L, "steps": a.steps,
           "ms_per_step": round(el / a.steps * 1e3, 4), "value_GBps": round(n * L / (el / a.steps) / 1e9, 2),
           "walk_kernel_ms_per_device": per_dev_ms,
           "async_ms_per_step": round(el_async / a.steps * 1e3, 4), "async_value_GBps": round(n * L / (el_async / a.steps) / 1e9, 2),
           "accepted_inputs": int(cnt), "bitmap_popcount_matches_count_on_every_checked_replica": bool(same)}
    node.close()
    print(json.dumps(out), flush=True)


LINE_LIMIT = 6000   # the driver keeps ~8 KB of stdout: the last line must stay well under it (tests/test_bench_line.py)
DETAIL_FILE = "bench_detail.json"


def _short_kernel(name, width=72):
    """rocprofv3's kernel name without the namespace, cut to `width` characters"""
    if not name:
        return name
    import re
    name = str(name).replace("fsmhip::", "").replace(" (mean length < 96 B, decided on the device)", "")
    # "(... decided / picked on the device[: why][, 1 of N launched])" -> "(picked on device[, 1 of N])"
    name = re.sub(r" \((?:mean length below the pick threshold, decided|decided|picked) on the device(?:: [^)]*?)?(?:, (1 of \d) launched)?\)",
                  lambda m: " (picked on device" + (", " + m.group(1) if m.group(1) else "") + ")", name)
    return name if len(name) <= width else name[:width - 1] + "~"


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d and d[k] is not None} if isinstance(d, dict) else d


def compact_sub(s):
    """One sub-result as the driver's line carries it: numbers only (the reference's own compact report,
    src/retest/reperf.c:804-954, is the model); everything else is in bench_detail.json."""
    roof = s.get("roofline") or {}
    alg = roof.get("algorithmic_bytes_per_launch")
    out = {"workload": s.get("workload"), "value": s.get("value"), "ms_per_step": s.get("ms_per_step"),
           "frac": roof.get("frac"), "kernel_ms": roof.get("kernel_ms_avg"), "kernel": _short_kernel(roof.get("kernel"), 56)}
    if roof.get("traffic") and alg:
        out["traffic_over_algorithmic"] = round(roof["traffic"] / alg, 4)
    if roof.get("frac_on_bytes_moved") is not None:
        out["frac_on_bytes_moved"] = roof["frac_on_bytes_moved"]
    for k in ("lds_chain_ceiling", "gather_ceiling"):
        c = roof.get(k)
        if isinstance(c, dict) and c.get("implied_GBps") is not None:
            out[k + "_GBps"] = c["implied_GBps"]
    par = s.get("parity_vs_cpu_sample") or s.get("parity_vs_main_run")
    if par is not None:
        out["parity"] = "bit-exact" if str(par).startswith("bit-exact") else "MISMATCH"
    if "full_parity" in s:
        out["full_parity_mismatches"] = s["full_parity"].get("mismatches")
        out["full_parity_rows"] = s["full_parity"].get("rows")
    cb = s.get("cpu_baseline")
    if isinstance(cb, dict) and cb.get("value") is not None:
        out["cpu_GBps"] = cb["value"]
    return {k: v for k, v in out.items() if v is not None or k in ("value",)}


def compact_line(res, detail_path=DETAIL_FILE, limit=LINE_LIMIT):
    """The ONE stdout line: headline + roofline + cpu_baseline + parity + one small entry per sub-result.
    The full record (every form, every sample description, every note) goes to `detail_path` and stderr."""
    out = {k: res.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                   "scaling", "vs_baseline", "dtype", "data")}
    cfg = res.get("config") or {}
    out["config"] = _pick(cfg, ("workload", "inputs_per_gpu", "input_len", "lines", "mean_len", "dfa_states", "byte_classes", "table_layout", "table_bytes",
                                "sharding", "accepted_inputs", "world_size"))
    if isinstance(out["config"].get("workload"), str) and len(out["config"]["workload"]) > 260:
        out["config"]["workload"] = out["config"]["workload"][:259] + "~"
    roof = res.get("roofline") or {}
    out["roofline"] = {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    if isinstance(out["roofline"]["bound"], str) and len(out["roofline"]["bound"]) > 8:
        out["roofline"]["bound"] = "hbm"
    out["roofline"].update(_pick(roof, ("kernel_ms_avg", "algorithmic_bytes_per_launch", "measured_read_stream_GBps", "frac_of_measured_stream", "frac_on_bytes_moved")))
    out["roofline"]["kernel"] = _short_kernel(roof.get("kernel"))
    if roof.get("traffic") and roof.get("algorithmic_bytes_per_launch"):
        out["roofline"]["traffic_over_algorithmic"] = round(roof["traffic"] / roof["algorithmic_bytes_per_launch"], 4)
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind", "sample", "fsm_exec_hoisted_value", "vm_v2_value", "vm_v2_allcores_value", "vm_v2_allcores_cores", "codegen_vmc_value"))
        if isinstance(c.get("sample"), str) and len(c["sample"]) > 150:
            c["sample"] = c["sample"][:149] + "~"
        out["cpu_baseline"] = c
    if "parity_vs_cpu_sample" in res:
        out["parity_vs_cpu_sample"] = res["parity_vs_cpu_sample"]
    if "full_parity" in res:
        out["full_parity"] = _pick(res["full_parity"], ("rows", "mismatches", "cpu_threads", "seconds"))
    if "multi_gpu" in res:
        out["multi_gpu"] = _pick(res["multi_gpu"], ("walk_kernel_ms_per_rank", "exchange_exposed_ms_per_step", "backend", "world_size", "rccl_version", "rccl_library"))
    if isinstance(res.get("node_front"), dict):
        out["node_front"] = _pick(res["node_front"], ("devices", "uses_rccl", "ms_per_step", "value_GBps", "async_ms_per_step", "async_value_GBps", "error"))
    if isinstance(res.get("multi_dfa"), dict):
        out["multi_dfa"] = _pick(res["multi_dfa"], ("dfas", "lines", "launches", "ms_per_call_multi", "ms_per_call_one_by_one", "speedup", "parity"))
    if isinstance(res.get("multi_dfa_bulk"), dict):
        out["multi_dfa_bulk"] = _pick(res["multi_dfa_bulk"], ("dfas", "lines_per_dfa", "launches", "ms_per_call", "ms_per_prepared_launch", "walked_GBps", "one_dfa_at_a_time_ms_extrapolated", "parity", "error"))
        if isinstance(res["multi_dfa_bulk"].get("roofline"), dict):
            out["multi_dfa_bulk"]["frac"] = res["multi_dfa_bulk"]["roofline"].get("frac")
    if res.get("sub_results"):
        out["sub_results"] = [compact_sub(s) for s in res["sub_results"]]
    out["detail"] = detail_path
    line = json.dumps(out, separators=(",", ":"))
    # never let the line outgrow the driver's window: shed the least important fields first
    for drop in ("kernel", "kernel_ms", "cpu_GBps", "full_parity_rows", "traffic_over_algorithmic"):
        if len(line) <= limit:
            break
        for s_ in out.get("sub_results", []):
            s_.pop(drop, None)
        line = json.dumps(out, separators=(",", ":"))
    if len(line) > limit:
        out.pop("node_front", None)
        out["config"].pop("workload", None)
        line = json.dumps(out, separators=(",", ":"))
    if len(line) > limit and out.get("sub_results"):
        out["sub_results"] = [_pick(s_, ("workload", "value", "frac", "parity")) for s_ in out["sub_results"]]
        line = json.dumps(out, separators=(",", ":"))
    return line


def progress(msg):
    """one line on stderr per leg (rank 0): a run that dies says where (stdout stays the one JSON line)"""
    if int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write("[bench %7.1f s] %s\n" % (time.time() - _T0, msg))
        sys.stderr.flush()


def emit(res):
    """Full record -> bench_detail.json (+ stderr); compact line -> stdout, last."""
    path = os.environ.get("FSM_BENCH_DETAIL", os.path.join(ROOT, DETAIL_FILE))
    try:
        with open(path, "w") as fh:
            json.dump(res, fh, indent=1)
            fh.write("\n")
        shown = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError as e:
        shown = f"(not written: {e})"
    print(json.dumps(res), file=sys.stderr, flush=True)
    sys.stdout.flush()
    print(compact_line(res, shown), flush=True)


WORKLOAD_TEXT = {
    "c2": "c2: BASELINE configs[1] -- PCRE [Ll]ibf+(sm)* DFA (5 states, absorbing accept), ",
    "c3": "c3: BASELINE configs[2] -- 1024 anchored PCRE unioned into one %d-state DFA, ",
    "c3t": ("c3t: the transition-dense twin of configs[2] -- 1024 patterns ^<pfx>([0-9][a-f])+(x|yz)$ unioned into one %d-state DFA, "
            "half the rows alternating digit / letter so that a live row changes state on EVERY byte, "),
    "c3u": ("c3u: the UNANCHORED (rx-style, src/rx/main.c:487-566) twin of configs[2] -- 1024 patterns <3-4 letters>[0-9]$ with the implicit "
            "leading .* unioned into one complete %d-state DFA (no DEAD default, a state change on almost every byte), rows random over [a-z0-9], "
            "every 2nd one ending in a pattern, "),
    "lds2": ("lds2: the pair-table regime -- 70 literals of 3-7 letters over a 12-letter alphabet, unanchored with end-ids (a dense "
             "mid-size DFA with few byte classes: one LDS lookup per TWO input bytes), uniform random text over that alphabet, "),
    "c5": "c5: BASELINE configs[4] -- Aho-Corasick DFA of %d literals (%d states, table > LDS), ",
}


T_START = time.perf_counter()


def over_budget():
    """The full-N checks of the SUB-results are the long part of a default run (host-side: every byte streamed back and walked by the
    CPU checker).  On a slow or throttled host they stop once the run is FSM_BENCH_TIME_BUDGET seconds old (default 420): the sample
    parity of every sub-result still runs, the main workload's full check always does, and the record says what was skipped."""
    return time.perf_counter() - T_START > float(os.environ.get("FSM_BENCH_TIME_BUDGET", "420"))


def main():
    a = parse()
    if a.node_front:
        import __graft_entry__ as ge
        ge.build()
        return node_front(a)
    import torch
    import torch.distributed as dist

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, rendezvous on 127.0.0.1)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("bench.py: --gpus %d without WORLD_SIZE: re-launching as %s" % (a.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # FSM_BENCH_FORCE_DIST=1: a ONE-rank run goes the N > 1 way all the same (RCCL communicator, the bitmap all-gather overlapped with
    # the next walk, the reductions, 1.25e8 inputs): the multi-GPU path on the real stack wherever only one GPU is at hand
    dist_on = world > 1 or bool(os.environ.get("FSM_BENCH_FORCE_DIST"))
    if world != a.gpus:   # the launcher's word wins; the line reports what ran
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: running {world} ranks", file=sys.stderr, flush=True)
    assert torch.cuda.is_available(), "bench.py needs a GPU: there is no CPU fallback"
    # FSM_BENCH_BACKEND=gloo lets the N > 1 plumbing be exercised on a box with fewer GPUs than ranks
    # (tests/test_gpu_parity.py::test_bench_two_ranks_share_one_gpu); the real runs use RCCL.
    backend = os.environ.get("FSM_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:     # forced: no launcher set the rendezvous up
            for k_, v_ in (("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
                os.environ.setdefault(k_, v_)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if dist_on:
        dist.barrier()
    import libfsm_amd as hip
    hip.load_library()
    from libfsm_amd.shard import shard_range

    L = a.len
    stream = torch.cuda.current_stream().cuda_stream

    # ---- sizes: per-GPU inputs of the main workload; one resident buffer serves every workload -------------
    def default_n(wl):
        if wl == "c5":
            return 10_000_000
        return 125_000_000 if dist_on else 100_000_000   # N > 1: configs[3] = 1e9 inputs over 8 GPUs

    requested = a.n if a.n > 0 else default_n(a.workload)
    n_total = requested if a.scaling == "strong" else requested * world
    n_total = n_total // (64 * world) * (64 * world) if dist_on else n_total   # equal shards of whole bitmap words
    first, n = shard_range(n_total, rank, world)
    free, _total = torch.cuda.mem_get_info()
    need = n * (L + 4) + n // 8 + (1 << 30)
    shrunk = False
    if need > free * 0.92:  # shrink rather than risk an OOM strike; reported in config
        n = int((free * 0.92 - (1 << 30)) // (L + 5)) // 64 * 64
        shrunk = True
        assert not dist_on, "the shard does not fit this GPU"
    buf_all = torch.empty((n, L), dtype=torch.uint8, device="cuda")
    end_all = torch.empty(n, dtype=torch.int32, device="cuda")

    def run(wl, variant=None, n_wl=None, first_wl=0, with_cpu=True):
        """Generate the workload's inputs in the resident buffer, time a.steps walks, return the result dict."""
        n_ = n if n_wl is None else min(n_wl, n)
        buf, end = buf_all[:n_], end_all[:n_]
        words = None
        if wl == "c5":
            words = c5_words(a.c5_words)
            flat = c5_flat(hip, a.c5_words)
        elif wl == "lds2":
            flat = lds2_flat(hip)
        else:
            flat = hip.FlatDfa.load(os.path.join(ROOT, "tests", "golden", "c1.npz" if wl == "c2" else wl + ".npz"))
        flags = a.layout | (hip.NO_EARLY_RETIRE if a.no_early_retire else 0)
        dfa = hip.HipDfa(flat, flags)
        for knob, v in ((hip.KNOB_INPUT_MODE, a.input_mode), (hip.KNOB_NB, a.nb), (hip.KNOB_WAVES, a.waves),
                        (hip.KNOB_BLOCKS_PER_CU, a.blocks_per_cu)):
            if v > 0 or (knob == hip.KNOB_INPUT_MODE and v >= 0):
                dfa.tune(knob, v)
        for kv in a.knob:
            k, v = kv.split("=")
            dfa.tune(int(k), int(v))
        if variant == "noskip":
            dfa.tune(KNOB_NOSKIP, 1)
        if variant == "loadskip":
            dfa.tune(hip.KNOB_EARLY_RETIRE, 3)   # bit 1: a lane in an absorbing state stops reading its row (fsm_exec's own early exit)
        info = dfa.info()
        nwords = (n_ + 63) // 64
        bm = torch.zeros(nwords, dtype=torch.int64, device="cuda")
        generate(hip, wl, buf.data_ptr(), n_, L, first_wl, words, buf)
        torch.cuda.synchronize()

        # N > 1: the all-gather of step k's bitmap runs on RCCL's stream while step k+1's walk kernel runs
        # (two bitmap / gather buffers); every gather is waited for before the timed region ends.
        bms = [bm, torch.zeros_like(bm)] if dist_on else [bm]
        gats = [torch.empty(nwords * world, dtype=torch.int64, device="cuda") for _ in range(2)] if dist_on else [None]
        pending = [None, None]
        tick = [0]
        kernel_ms = []

        def step(record):
            k = tick[0] % len(bms)
            tick[0] += 1
            if dist_on and pending[k] is not None:
                pending[k].wait()      # stream-level: the gather that last read this bitmap buffer is done
                pending[k] = None
            dfa.exec_batch_device(buf.data_ptr(), L, n_, end.data_ptr(), bms[k].data_ptr(), stream=stream)
            if record:
                kernel_ms.append(dfa.last_kernel_ms())  # HIP events on the launch stream, around the walk kernel only
            if dist_on:
                # the match bitmap over RCCL/xGMI (libfsm_amd/shard.py)
                pending[k] = dist.all_gather_into_tensor(gats[k], bms[k], async_op=True)

        def drain():
            for k in range(len(pending)):
                if pending[k] is not None:
                    pending[k].wait()
                    pending[k] = None

        for _ in range(4):  # setup, untimed: the first launches after a long generator kernel run at ramping clocks
            dfa.exec_batch_device(buf.data_ptr(), L, n_, end.data_ptr(), bm.data_ptr(), stream=stream)
        kernel_name = dfa.last_kernel_name()   # the library's own word for what it launched (as rocprofv3 lists it)
        for _ in range(a.warmup):
            step(False)
        drain()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step(True)
        drain()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        if dist_on:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

        acc_t = (end != -1).sum().to(torch.int64).reshape(1)
        rank_kms = None
        if dist_on:
            dist.all_reduce(acc_t)
            mine = torch.tensor([float(np.mean(kernel_ms))], dtype=torch.float64, device="cuda")
            allk = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allk, mine)
            rank_kms = [round(float(t.item()), 4) for t in allk]
        if rank != 0:
            dfa.close()
            return None

        ms_step = elapsed / a.steps * 1e3
        value = float(n_) * L * world / (elapsed / a.steps) / 1e9
        k_ms = float(np.mean(kernel_ms))
        alg_bytes = float(n_) * (L + 4)  # L bytes read + 4 B end state written per input
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        # HBM traffic per launch from the PMC counters: collected in separate rocprofv3 passes (tools/profile.sh),
        # never inside this run -- quoted only when the recorded launch is the same workload, size and kernel
        traffic, traffic_source = None, None
        pj = os.path.join(ROOT, "profiles", f"pmc_{wl}.json" if variant is None else f"pmc_{wl}_{variant}.json")
        if variant in (None, "loadskip") and os.path.exists(pj):
            try:
                t = json.load(open(pj))
                if int(t.get("n", 0)) == n_ and int(t.get("len", 0)) == L and t.get("kernels_sha16") == kernels_sha16():
                    traffic = t.get("hbm_bytes_per_launch")
                    traffic_source = (f"profiles/{t.get('source')}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over this "
                                      f"workload at this size (kernel {str(t.get('kernel'))[:60]}); recorded, not measured in this run")
            except Exception:
                traffic = None
        text = WORKLOAD_TEXT[wl] % ((flat.nstates,) if wl in ("c3", "c3t", "c3u") else (len(words), flat.nstates) if wl == "c5" else ())
        if variant == "noskip":
            text += "chunk skip disabled (every byte pays its self-loop test: the transition-dense bound of this table), "
        if variant == "loadskip":
            text += ("per-lane load skip ON (a lane whose input can no longer change state stops reading it, as fsm_exec stops pulling bytes "
                     "at a missing edge: the rate counts the bytes MATCHED, fewer are touched -- roofline.early_retire), ")
        # a kernel that does not fetch every byte (per-lane load skip) is priced on the bytes it TOUCHES; its rate over all
        # the bytes it matched is reported beside it, under its own name
        roof_bytes = alg_bytes
        if variant == "loadskip":
            roof_bytes = traffic if traffic else None
        res = {
            "value": round(value, 2), "ms_per_step": round(ms_step, 4),
            "config": {
                "workload": text + f"{n_} x {L} B synthetic inputs per GPU resident in HBM, 64 inputs/wavefront",
                "inputs_per_gpu": n_, "input_len": L, "dfa_states": flat.nstates, "byte_classes": info["nclasses"],
                "table_layout": info["layout_name"], "table_bytes": info["table_bytes"], "lds_bytes_per_block": info["lds_bytes"],
                "waves_per_block": info["waves_per_block"],
                "sharding": (f"{world} contiguous index ranges ({a.scaling} scaling); RCCL all-gather of each step's accept bitmap, overlapped with the next step's walk"
                             if dist_on else "single GPU"),
                "accepted_inputs": int(acc_t.item()),
            },
            **({"multi_gpu": {"world_size": dist.get_world_size(), "backend": dist.get_backend(), **rccl_identity(torch), "walk_kernel_ms_per_rank": rank_kms,
                              "exchange_exposed_ms_per_step": round(max(0.0, ms_step - max(rank_kms)), 4),
                              "note": "ms_per_step minus the slowest rank's walk kernel: what the RCCL all-gather of the accept bitmap (overlapped with the next step's walk) and the host loop add"}}
               if rank_kms else {}),
            "roofline": {"bound": "hbm" if wl != "c5" else "instruction issue of a table walk whose table is in L2 (HBM figures for uniformity; see gather_ceiling)",
                         "achieved": None if roof_bytes is None else round(roof_bytes / (k_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": None if roof_bytes is None else round(roof_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_name, "kernel_ms_avg": round(k_ms, 4),
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        # "also report touched bytes when early-retire is enabled".  A wavefront stops reading once all 64 of its
        # inputs sit in absorbing states (fsm_exec's own early exit, exec.c:133-138); what was actually fetched is the
        # read side of the recorded PMC traffic.
        res["roofline"]["early_retire"] = {
            "enabled": True,   # the library default; FSM_HIP_NO_EARLY_RETIRE at create time switches it off
            "per_lane_load_skip": variant == "loadskip",
            "matched_GBps": round(achieved, 2),   # algorithmic bytes (every input byte + 4) over the kernel time: what the caller sees
            "touched_bytes_per_launch": traffic,
            "touched_over_algorithmic": round(traffic / alg_bytes, 4) if traffic else None,
            "note": "HBM bytes per launch from the recorded PMC passes (reads + the 4-byte results): ~1.00 means every input byte was still fetched",
        }
        if wl == "c5":
            res["roofline"]["note"] = ("round 4: the lazy walk (walk_lazy.h) enters a state beyond the LDS set without fetching its record -- 0.086 L2 requests per "
                                       "input byte where the record-as-state walk made 0.33 -- and is bound by instruction issue now, not by gathers and not by HBM "
                                       "(profiles/r06b_c5_lazy_pmc_rows2.txt; DESIGN.md section 3): the fraction of HBM peak is reported for uniformity, "
                                       "roofline.gather_ceiling gives the memory-side denominator")
        if not dist_on and with_cpu and not a.no_cpu_baseline and a.cpu_sample != 0:
            sample = a.cpu_sample if a.cpu_sample > 0 else (400_000 if wl == "c2" else 100_000)
            idx = sample_indices(n_, sample)
            tidx = torch.from_numpy(idx).cuda()
            rows = buf[tidx].cpu().numpy()                              # the bytes the GPU walked
            cb, parity = cpu_baseline(hip, wl, flat, L, rows, end[tidx].cpu().numpy().view(np.uint32), words)
            # the device rows are what the host generator produces for the same indices (spot check on a run)
            lo = int(idx[len(idx) // 2])
            cnt = min(256, n_ - lo)
            twin = bool(np.array_equal(buf[lo:lo + cnt].cpu().numpy(), generate_host(hip, wl, cnt, L, first_wl + lo, words)))
            res["cpu_baseline"] = cb
            res["parity_vs_cpu_sample"] = "bit-exact" if parity and twin else "MISMATCH"
            res["parity_sample"] = f"{len(idx)} inputs: seeded stratified sample of [0, {n_}) + first/last 64 rows" + (" + rows around byte offset 2^32" if n_ > (1 << 22) + 64 else "")
            if not (parity and twin):
                res["value"] = None  # a fast wrong answer is not a result
        # the full-N compare, for the main workload of a default run -- and for c5 wherever it runs (round 4: its
        # table is 1 GB as the oracle keeps it and the cpu_baseline leg has built it already; 1e7 rows take the walker's threads
        # about a minute)
        if not dist_on and variant is None and not a.no_full_parity and with_cpu and not a.no_cpu_baseline and (a.full_parity or wl == a.workload or wl == "c5") \
                and wl != a.workload and over_budget():
            res["full_parity_skipped"] = "time budget (FSM_BENCH_TIME_BUDGET) reached: sample parity only"
        elif not dist_on and variant is None and not a.no_full_parity and with_cpu and not a.no_cpu_baseline and (a.full_parity or wl == a.workload or wl == "c5"):
            res["full_parity"] = full_parity(torch, flat, buf, end, n_, L)
            if res["full_parity"]["mismatches"]:
                res["value"] = None
        if wl == "c5":
            # a denominator for this walk: what the memory system gives 16-byte gathers from a table of this size at full occupancy
            # (the walk's own-record gathers, 0.07 per input byte), and what the instruction stream allows
            try:
                ng = 1 << 28
                gms = hip.gather_probe_ms(buf.data_ptr(), 18 << 20, ng, 16, bm.data_ptr(), stream)
                per_byte = 0.068
                res["roofline"]["gather_ceiling"] = {
                    "probe": "fsm_hip_gather_probe_ms: 2^28 independent 16-byte gathers over an 18 MB window, 8 workgroups of 256 per CU",
                    "gathers_per_second": round(ng / gms * 1e3, 0), "walk_gathers_per_input_byte": per_byte,
                    "walk_gathers_source": "profiles/r06b_c5_lazy_pmc_rows2.txt: TCP_TCC_READ_REQ 0.086 per input byte, 0.016 of them the input's own lines",
                    "implied_GBps": round(ng / gms * 1e3 / per_byte / 1e9, 1),
                    "note": "the walk is not at this ceiling: with 25 vector instructions per input byte-step it is bound by instruction issue "
                            "(VALU 55-60 % busy, LDS 51 %, TA 56 %: profiles/r06b_c5_lazy_pmc_rows2.txt); the issue ceiling at 4 cycles per wave64 "
                            "instruction is ~1.3 TB/s"}
            except Exception as e:  # noqa: BLE001
                res["roofline"]["gather_ceiling"] = {"error": repr(e)[:200]}
        res["_buf"] = (buf, bm)
        # a digest of all n end states: a variant run over the same inputs (noskip, loadskip) must reproduce the main run's
        res["_digest"] = (int(end.to(torch.int64).sum().item()), int((end.to(torch.int64) * (torch.arange(n_, device="cuda", dtype=torch.int64) % 1000003 + 1)).sum().item()),
                          int(acc_t.item())) if not dist_on else None
        dfa.close()
        return res

    def run_lines(wl, kind):
        """The front retest / rx actually drive (src/retest/main.c:1114, src/retest/reperf.c:772-784): short, packed lines.
        Input i = the first len[i] bytes of row i of workload `wl` (so a pattern row keeps its prefix and stays alive), packed
        back to back; `short`: 1e8 lines of 8..64 bytes, `ragged`: 2e7 lines of 0..1024 bytes -- both >= 3.6 GB walked per
        launch.  Timed in four metadata forms; the line's `value` is the u64-offsets + u32-end-state form and its roofline
        counts every byte that form moves: sum(len) + 8 B of offsets + 4 B of result per line."""
        lo, hi, n_l = (8, 64, 100_000_000) if kind == "short" else (0, 1024, 20_000_000)
        n_l = min(n_l, n)
        words = None
        if wl == "c5":
            words = c5_words(a.c5_words)
            flat = c5_flat(hip, a.c5_words)
        else:
            flat = hip.FlatDfa.load(os.path.join(ROOT, "tests", "golden", "c1.npz" if wl == "c2" else wl + ".npz"))
        dfa = hip.HipDfa(flat, a.layout)
        for kv in a.knob:
            k_, v_ = kv.split("=")
            dfa.tune(int(k_), int(v_))
        rows = buf_all[:n_l]
        if wl == "c5":     # rows over the literals' alphabet (the literals are planted at the LINES' ends, below)
            hip.gen_inputs_device(rows.data_ptr(), n_l, L, 0, SEED, ALPHA64)
        else:
            generate(hip, wl, rows.data_ptr(), n_l, L, 0)
        g = torch.Generator(device="cuda").manual_seed(SEED & 0x7FFFFFFF)
        lens = torch.randint(lo, hi + 1, (n_l,), device="cuda", dtype=torch.int32, generator=g)
        off = torch.zeros(n_l + 1, dtype=torch.int64, device="cuda")
        torch.cumsum(lens, 0, dtype=torch.int64, out=off[1:])
        total = int(off[-1].item())
        packed = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
        hip.gen_pack_rows_device(rows.data_ptr(), L, lens.data_ptr(), off.data_ptr(), n_l, hi, packed.data_ptr())
        if wl == "c5":
            # every 8th line ends with the literal (index * 2654435761) % nwords, where it is long enough: accepted by the
            # right-anchored automaton
            Wt, wl_len = c5_tail_table(words)
            Wd, ld = torch.from_numpy(Wt).cuda(), torch.from_numpy(wl_len).cuda()
            idx = torch.arange(0, n_l, 8, device="cuda")
            widx = (idx * 2654435761) % len(words)
            for l in sorted(set(wl_len.tolist())):
                m = (ld[widx] == l) & (lens[idx].to(torch.int64) >= l)
                if bool(m.any()):
                    pos = (off[idx[m] + 1] - l).unsqueeze(1) + torch.arange(l, device="cuda").unsqueeze(0)
                    packed[pos.reshape(-1)] = Wd[widx[m], :l].reshape(-1)
            del Wd, ld, idx, widx
        off32 = off.to(torch.int32) if total < (1 << 32) else None
        end = end_all[:n_l]
        end2 = torch.empty_like(end)
        bm = torch.zeros((n_l + 63) // 64, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        forms = {}

        def timed(name, call, meta_bytes, out_bytes):
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            ms = []
            t0 = time.perf_counter()
            for _ in range(a.steps):
                call()
                ms.append(dfa.last_kernel_ms())
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / a.steps * 1e3
            k_ms = float(np.mean(ms))
            tot = total + n_l * (meta_bytes + out_bytes)
            forms[name] = {"ms_per_step": round(wall, 4), "kernel_ms_avg": round(k_ms, 4), "walked_GBps": round(total / wall / 1e6, 2),
                           "bytes_per_launch": tot, "metadata_bytes_per_line": meta_bytes, "result_bytes_per_line": out_bytes,
                           "total_GBps": round(tot / k_ms / 1e6, 2), "frac_of_hbm_peak": round(tot / k_ms / 1e6 / HBM_PEAK_GBS, 4),
                           "kernel": dfa.last_kernel_name()}
            return wall, k_ms

        wall, k_ms = timed("off64_end", lambda: dfa.exec_batch_offsets_device(packed.data_ptr(), off.data_ptr(), n_l, end.data_ptr(), 0, stream=stream), 8, 4)
        ok_forms = True
        only64 = os.environ.get("FSM_BENCH_LINES_FORMS") == "off64"      # profiling runs: one kernel, one form
        if off32 is not None and not only64:
            timed("off32_end", lambda: dfa.exec_batch_offsets32_device(packed.data_ptr(), off32.data_ptr(), n_l, end2.data_ptr(), 0, stream=stream), 4, 4)
            ok_forms = ok_forms and bool(torch.equal(end, end2))
        end2.fill_(7)
        # lengths only: the pre-pass reads the lengths once more (4 B) and writes 8 B per 64 lines; its time is inside ms_per_step, not kernel_ms
        if not only64:
            timed("len_end", lambda: dfa.exec_batch_lengths_device(packed.data_ptr(), lens.data_ptr(), n_l, end2.data_ptr(), 0, stream=stream), 8.125, 4)
            ok_forms = ok_forms and bool(torch.equal(end, end2))
            timed("len_bitmap", lambda: dfa.exec_batch_lengths_device(packed.data_ptr(), lens.data_ptr(), n_l, 0, bm.data_ptr(), stream=stream), 8.125, 0.125)
        acc = int((end != -1).sum().item())
        ok_forms = ok_forms and (only64 or bool(np.array_equal(np.unpackbits(bm.cpu().numpy().view(np.uint8), bitorder="little")[:n_l].astype(bool), (end != -1).cpu().numpy())))
        info = dfa.info()
        f0 = forms["off64_end"]
        res = {"workload": f"{wl}_{kind}", "value": f0["walked_GBps"], "unit": "GB/s of line bytes walked", "ms_per_step": f0["ms_per_step"],
               "config": {"workload": f"{kind} lines on the {wl} table: input i = the first len[i] bytes of row i of the {wl} workload, len uniform in [{lo}, {hi}], "
                                      f"{n_l} lines packed back to back ({total} B) resident in HBM; u64 offsets in, u32 end states out",
                          "lines": n_l, "line_bytes": total, "mean_len": round(total / n_l, 2), "dfa_states": flat.nstates, "table_layout": info["layout_name"],
                          "accepted_inputs": acc},
               "roofline": {"bound": "hbm", "achieved": f0["total_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": f0["frac_of_hbm_peak"],
                            "kernel": f0["kernel"], "kernel_ms_avg": f0["kernel_ms_avg"], "algorithmic_bytes_per_launch": f0["bytes_per_launch"],
                            "algorithmic_bytes": "sum(len) + 8 B offsets + 4 B end state per line", "traffic": None, "traffic_source": None},
               "forms": forms}
        pj = os.path.join(ROOT, "profiles", f"pmc_{wl}_{kind}.json")
        if os.path.exists(pj):
            try:
                t = json.load(open(pj))
                if int(t.get("n", 0)) == n_l and t.get("kernels_sha16") == kernels_sha16():
                    res["roofline"]["traffic"] = t.get("hbm_bytes_per_launch")
                    res["roofline"]["traffic_source"] = (f"profiles/{t.get('source')}: rocprofv3 --pmc passes over this sub-result -- reads from TCC_EA0_RDREQ_32B / _64B / _128B "
                                                         "(32 a + 64 b + 128 c bytes), writes from WRITE_SIZE (recorded, not measured in this run)")
                    # a kernel that stops reading lines which can no longer change state moves fewer bytes than the caller handed it:
                    # its rate on the bytes it MOVED beside the one on the bytes it matched
                    if res["roofline"]["traffic"]:
                        res["roofline"]["frac_on_bytes_moved"] = round(res["roofline"]["traffic"] / f0["kernel_ms_avg"] / 1e6 / HBM_PEAK_GBS, 4)
            except Exception:
                pass
        # parity: a stratified sample of the lines -- their bytes gathered from the PACKED buffer the kernel walked -- against
        # the oracle's walk
        if not a.no_cpu_baseline and a.cpu_sample != 0:
            idx = sample_indices(n_l, 100_000)
            tidx = torch.from_numpy(idx).cuda()
            sl = lens[tidx].to(torch.int64)
            so = torch.zeros(len(idx) + 1, dtype=torch.int64, device="cuda")
            torch.cumsum(sl, 0, out=so[1:])
            stot = int(so[-1].item())
            src = torch.repeat_interleave(off[tidx] - so[:-1], sl) + torch.arange(stot, device="cuda") if stot else torch.zeros(0, dtype=torch.int64, device="cuda")
            sbase = packed[src].cpu().numpy() if stot else np.zeros(1, np.uint8)
            o = get_oracle(flat)
            want = o.table_walk_packed_mt(sbase, so.cpu().numpy().astype(np.uint64), 1)
            got = end[tidx].cpu().numpy().view(np.uint32)
            del src
            res["cpu_baseline"] = {"kind": "port", "value": round(stot / 1e9 / max(o.last_seconds, 1e-9), 5), "unit": "GB/s", "cores": 1,
                                   "sample": f"oracle dense-table walker (oracle/dfa_oracle.c), 1 thread, {len(idx)} of the lines"}
            res["parity_vs_cpu_sample"] = "bit-exact" if np.array_equal(got, want) and ok_forms else "MISMATCH"
            res["parity_sample"] = f"{len(idx)} lines: seeded stratified sample + first/last 64, bytes taken from the packed buffer; the other metadata forms reproduce all {n_l} end states"
            if res["parity_vs_cpu_sample"] != "bit-exact":
                res["value"] = None
            # ... and EVERY line: the packed buffer streams back in slices of whole lines (<= 2 GiB each) and the oracle walks
            # them on all granted host cores
            if not a.no_full_parity and over_budget():
                res["full_parity_skipped"] = "time budget (FSM_BENCH_TIME_BUDGET) reached: sample parity only"
            elif not a.no_full_parity:
                ncores, _ = host_cores()
                offh = off.cpu().numpy().astype(np.uint64)
                bad, t0, t_cpu, r0 = 0, time.perf_counter(), 0.0, 0
                while r0 < n_l:
                    r1 = int(np.searchsorted(offh, offh[r0] + np.uint64(2 << 30), side="right")) - 1
                    r1 = min(n_l, max(r1, r0 + 1))
                    b0, b1 = int(offh[r0]), int(offh[r1])
                    chunk = packed[b0:b1].cpu().numpy() if b1 > b0 else np.zeros(1, np.uint8)
                    want_all = o.table_walk_packed_mt(chunk, offh[r0:r1 + 1] - offh[r0], ncores)
                    t_cpu += o.last_seconds
                    bad += int((end[r0:r1].cpu().numpy().view(np.uint32) != want_all).sum())
                    r0 = r1
                res["full_parity"] = {"rows": n_l, "mismatches": bad, "cpu_threads": ncores, "seconds": round(time.perf_counter() - t0, 1),
                                      "cpu_walk_GBps": round(total / 1e9 / max(t_cpu, 1e-9), 2), "checker": "oracle dense-table walker (oracle/dfa_oracle.c), all lines, from the packed buffer"}
                if bad:
                    res["value"] = None
        dfa.close()
        del packed, off, lens, end2, bm
        return res

    def run_eager40():
        """SURVEY.md 8(f)2 / exec.c:126-144: the eager-output walk on the kind of DFA it exists for -- the reference's
        fsm_union_repeated_pattern_group over 40 unanchored literals, one eager id each (tests/golden/bench/eager40.npz, frozen
        from the real reference by tests/golden/make_eager40.py); 1e7 x 1 KiB rows of lowercase text, a pattern planted in
        every 4th.  Per input: L bytes read + 4 B end state + 8 B id set written."""
        z = np.load(os.path.join(ROOT, "tests", "golden", "bench", "eager40.npz"))
        flat = hip.FlatDfa.load(z)
        words = bytes(z["patterns"]).split(b"\n")
        n_e = min(10_000_000, n)
        buf, end = buf_all[:n_e], end_all[:n_e]
        alpha = b"abcdefghijklmnopqrstuvwxyz"
        hip.gen_inputs_device(buf.data_ptr(), n_e, L, 0, 7, alpha, words[0], 4)
        dfa = hip.HipDfa(flat, a.layout)
        W = dfa.eager_words()
        sets = torch.zeros((n_e, W), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        res = {"workload": "c3_eager40"}
        for name, fn in (("plain", lambda: dfa.exec_batch_device(buf.data_ptr(), L, n_e, end.data_ptr(), 0, stream=stream)),
                         ("eager", lambda: dfa.exec_batch_eager_device(buf.data_ptr(), L, n_e, end.data_ptr(), sets.data_ptr(), stream=stream))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ms = []
            t0 = time.perf_counter()
            for _ in range(a.steps):
                fn()
                ms.append(dfa.last_kernel_ms())
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / a.steps * 1e3
            res[name] = {"ms_per_step": round(wall, 4), "kernel_ms_avg": round(float(np.mean(ms)), 4), "GBps": round(n_e * L / wall / 1e6, 2), "kernel": dfa.last_kernel_name()}
        k_ms = res["eager"]["kernel_ms_avg"]
        alg = float(n_e) * (L + 4 + 8 * W)
        info = dfa.info()
        res.update(value=res["eager"]["GBps"], ms_per_step=res["eager"]["ms_per_step"],
                   config={"workload": f"eager outputs: fsm_union_repeated_pattern_group over 40 unanchored literals ({flat.nstates} states, {dfa.eager_id_count()} eager ids), "
                                       f"{n_e} x {L} B lowercase rows, a pattern planted in every 4th; end state + id set per input",
                           "inputs_per_gpu": n_e, "input_len": L, "dfa_states": flat.nstates, "table_layout": info["layout_name"], "eager_ids": dfa.eager_id_count(),
                           "table_bytes": info["table_bytes"], "lds_bytes_per_block": info["lds_bytes"], "waves_per_block": info["waves_per_block"]},
                   roofline={"bound": "hbm", "achieved": round(alg / k_ms / 1e6, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / k_ms / 1e6 / HBM_PEAK_GBS, 4),
                             "kernel": res["eager"]["kernel"], "kernel_ms_avg": k_ms, "algorithmic_bytes_per_launch": alg,
                             "algorithmic_bytes": "L bytes + 4 B end state + 8 B id set per input", "traffic": None, "traffic_source": None,
                             "plain_walk_same_dfa_GBps": res["plain"]["GBps"]})
        pj = os.path.join(ROOT, "profiles", "pmc_c3_eager40.json")
        if os.path.exists(pj):
            try:
                t = json.load(open(pj))
                if int(t.get("n", 0)) == n_e and t.get("kernels_sha16") == kernels_sha16():
                    res["roofline"]["traffic"] = t.get("hbm_bytes_per_launch")
                    res["roofline"]["traffic_source"] = f"profiles/{t.get('source')}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this sub-result (recorded, not measured in this run)"
            except Exception:
                pass
        if not a.no_cpu_baseline and a.cpu_sample != 0:
            # the oracle's eager walk (exec.c:126-144 restated) on a stratified sample: end state AND id set per input
            o = get_oracle(flat)
            idx = sample_indices(n_e, 20_000)
            tidx = torch.from_numpy(idx).cuda()
            rows = buf[tidx].cpu().numpy()
            t0 = time.perf_counter()
            _, wend, wsets = o.exec_eager(rows, None, cap=dfa.eager_id_count() + 8)
            t_cpu = time.perf_counter() - t0
            ids = np.array([dfa.eager_id(b) for b in range(dfa.eager_id_count())], np.uint32)
            got_end = end[tidx].cpu().numpy().view(np.uint32)
            bits = np.unpackbits(sets[tidx].cpu().numpy().view(np.uint8).reshape(len(idx), W * 8), axis=1, bitorder="little")[:, :len(ids)].astype(bool)
            ok = bool(np.array_equal(got_end, wend)) and all(np.array_equal(ids[bits[i]], np.sort(np.asarray(wsets[i], np.uint32))) for i in range(len(idx)))
            res["cpu_baseline"] = {"kind": "port", "value": round(rows.size / 1e9 / t_cpu, 5), "unit": "GB/s", "cores": 1,
                                   "sample": f"oracle eager walk (oracle/dfa_oracle.c, exec.c:126-144 restated), 1 thread, {len(idx)} inputs"}
            res["parity_vs_cpu_sample"] = "bit-exact" if ok else "MISMATCH"
            res["parity_sample"] = f"{len(idx)} inputs (stratified): end state and eager id set; {int((sets != 0).any(dim=1).sum().item())} of {n_e} inputs fired an id"
            if not ok:
                res["value"] = None
        dfa.close()
        return res

    def lds_chain_ceiling(sub):
        """roofline.lds_chain_ceiling for a lookup-layout sub-result: the same dependent chain (random ds_read_b32 per byte, extract,
        add, compare, select) with the inputs in registers, at the sub-result's table size and wavefront shape."""
        try:
            cfg = sub["config"]
            tb = int(max(2048, min(160 * 1024 - 1024, cfg.get("table_bytes") or cfg["lds_bytes_per_block"])))
            waves = int(cfg["waves_per_block"])
            bpc = max(1, min((160 * 1024) // max(tb, 1), 32 // max(waves, 1)))
            scratch = torch.zeros(4, dtype=torch.int32, device="cuda")
            g = hip.lds_chain_probe_gbps(tb, waves, bpc, 1 << 16, scratch.data_ptr(), stream)
            sub["roofline"]["lds_chain_ceiling"] = {
                "probe": f"fsm_hip_lds_chain_probe_gbps: {waves} wavefronts x {bpc} workgroup(s) per CU beside a {tb} B random table, 65 536 dependent "
                         "byte-steps per lane (ds_read_b32 -> bfe -> add -> compare -> select), input bytes made in registers",
                "implied_GBps": round(g, 1), "achieved_over_ceiling": round(sub["roofline"]["achieved"] / g, 4) if g > 0 and sub["roofline"].get("achieved") else None}
        except Exception as e:  # noqa: BLE001
            sub["roofline"]["lds_chain_ceiling"] = {"error": repr(e)[:200]}

    def run_multi_dfa():
        """north_star's many-DFA batches on the reference's own retest corpus (37 automata / 115 lines, tests/golden/retest): per
        pass over the 37 records, table builds included, one by one (fsm_hip_dfa_create + fsm_hip_exec_batch_offsets each) against
        ONE fsm_hip_exec_multi over dfas created with FSM_HIP_DEFER_UPLOAD; every end state against the frozen fsm_exec answers."""
        import glob
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from common import Golden
        gs = [Golden(p) for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "retest", "*.npz")))]
        jobs = [g.strings() for g in gs]
        want = [np.where(g.ret == 1, g.end, 0xFFFFFFFF).astype(np.uint32) for g in gs]

        def one_by_one():
            outs = []
            for g, j in zip(gs, jobs):
                d = hip.HipDfa(g.flat)
                outs.append(d.exec_strings(j)[0])
                d.close()
            return outs

        def multi():
            ds = [hip.HipDfa(g.flat, hip.DEFER_UPLOAD) for g in gs]
            outs = [e for e, _ in hip.exec_multi(ds, jobs)]
            for d in ds:
                d.close()
            return outs

        ok = all(np.array_equal(e, w) for e, w in zip(multi(), want)) and all(np.array_equal(e, w) for e, w in zip(one_by_one(), want))
        launches = hip.multi_last_launches()
        t = {}
        for name, fn in (("one_by_one", one_by_one), ("multi", multi)):
            fn()
            ts = []
            for _ in range(10):
                t0 = time.perf_counter()
                fn()
                ts.append((time.perf_counter() - t0) * 1e3)
            t[name] = float(np.median(ts))
        return {"dfas": len(gs), "lines": sum(len(j) for j in jobs), "launches": launches, "ms_per_call_multi": round(t["multi"], 3),
                "ms_per_call_one_by_one": round(t["one_by_one"], 3), "speedup": round(t["one_by_one"] / t["multi"], 1),
                "parity": "bit-exact" if ok else "MISMATCH",
                "note": "wall time of a pass over all records, fsm_hip_dfa_create included (retest builds a DFA per record, src/retest/main.c:1056-1058); "
                        "end states against the reference's frozen fsm_exec answers"}

    def run_multi_dfa_bulk():
        """The many-DFA front at THROUGHPUT size: 1 024 automata (the 37 retest goldens, cycled) x 100 000 lines of 64 bytes each,
        device pointers, ONE fsm_hip_exec_multi_device -- one launch whose workgroups of four wavefronts map to (dfa, 256 lines)
        and share that dfa's table copy in LDS.  Beside it: the same jobs one dfa at a time (fsm_hip_exec_batch_offsets_device,
        tables uploaded beforehand), which is also the parity check on a sample of the jobs."""
        import glob
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from common import Golden
        gs = [Golden(p) for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "retest", "*.npz")))]
        K, nl, ll = 1024, 100_000, 64
        text = buf_all.view(-1)[: K * nl * ll]
        hip.gen_inputs_device(text.data_ptr(), K * nl, ll, 0, SEED ^ 0x77, bytes(range(32, 127)))
        off = (torch.arange(nl + 1, device="cuda", dtype=torch.int64) * ll)       # every job: 100 000 lines of 64 bytes, its own slice of the text
        ends = torch.empty(K * nl, dtype=torch.int32, device="cuda")      # (its own block: the main workload's end states are 1e8, these 1.024e8)
        ds = [hip.HipDfa(gs[q % len(gs)].flat, hip.DEFER_UPLOAD) for q in range(K)]
        jobs = [(text.data_ptr() + q * nl * ll, off.data_ptr(), nl, ends.data_ptr() + q * nl * 4, 0) for q in range(K)]
        torch.cuda.synchronize()

        def once():
            hip.exec_multi_device(ds, jobs, stream=stream)
        for _ in range(2):
            once()
        torch.cuda.synchronize()
        launches, fused = hip.multi_last_launches(), hip.multi_last_fused_jobs()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            once()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        # the same submission PREPARED (fsm_hip_multi_prepare: descriptors, tile map and tables on the device once): a call is the kernel alone
        pr = hip.MultiPrepared(ds, [j + (0,) for j in jobs], 1)
        for _ in range(2):
            pr.launch(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            pr.launch(stream)
        torch.cuda.synchronize()
        ms_p = (time.perf_counter() - t0) / reps * 1e3
        pr.close()
        # parity + the one-by-one figure on every 64th job (its own dfa, its own planned layout)
        sample = list(range(0, K, 64))
        ok, t_one = True, 0.0
        chk = torch.empty(nl, dtype=torch.int32, device="cuda")
        for q in sample:
            one = hip.HipDfa(gs[q % len(gs)].flat)
            one.exec_batch_offsets_device(jobs[q][0], off.data_ptr(), nl, chk.data_ptr(), 0, stream=stream)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            one.exec_batch_offsets_device(jobs[q][0], off.data_ptr(), nl, chk.data_ptr(), 0, stream=stream)
            torch.cuda.synchronize()
            t_one += time.perf_counter() - t1
            ok = ok and bool(torch.equal(chk, ends[q * nl:(q + 1) * nl]))
            one.close()
        for d in ds:
            d.close()
        nbytes = K * nl * (ll + 8 + 4)
        return {"dfas": K, "lines_per_dfa": nl, "line_bytes": ll, "launches": launches, "fused_jobs": fused, "ms_per_call": round(ms, 3),
                "ms_per_prepared_launch": round(ms_p, 3), "walked_GBps": round(K * nl * ll / ms_p / 1e6, 1),
                "roofline": {"bound": "hbm", "achieved": round(nbytes / ms_p / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nbytes / ms_p / 1e6 / HBM_PEAK_GBS, 4),
                             "algorithmic_bytes": "64 B of text + 8 B offset + 4 B end state per line",
                             "timed": "the prepared launch (one kernel, host clock around 5 of them); ms_per_call = fsm_hip_exec_multi_device, which also builds and "
                                      "copies 1 024 descriptors and tables per call"},
                "one_dfa_at_a_time_ms_extrapolated": round(t_one / len(sample) * K * 1e3, 1),
                "parity": ("bit-exact" if ok else "MISMATCH") + f" ({len(sample)} of the jobs against their dfa's own walk of the same lines)"}

    if "_" in a.workload:       # the packed-lines front alone
        wl, kind = a.workload.split("_")
        r = run_eager40() if kind == "eager40" else run_lines(wl, kind)
        r["config"].setdefault("lines", r["config"].get("inputs_per_gpu"))
        r["config"].setdefault("mean_len", r["config"].get("input_len"))
        r.update(metric="input GB/s matched (whole node)", n_gpus=world, steps=a.steps, warmup=a.warmup, higher_is_better=True, scaling=a.scaling,
                 vs_baseline=None, dtype="u8", data="synthetic")
        r["config"]["inputs_per_gpu"] = r["config"]["lines"]
        r["config"]["input_len"] = r["config"]["mean_len"]
        emit(r)
        return
    progress("main workload %s" % a.workload)
    main_res = run(a.workload)
    subs = []
    if not dist_on and a.subs == "auto" and a.n == 0 and not shrunk:
        plan = []
        if a.workload == "c3":
            plan.append(("c3", "noskip", None))
            plan.append(("c3", "loadskip", None))
        for wl in ("c3", "c3t", "c3u", "lds2", "c2", "c5"):
            if wl != a.workload:
                plan.append((wl, None, default_n(wl)))
        # the short / packed front of retest and rx, at steady-state size, on the C2 and C3 tables
        for wl in ("c2", "c3", "c5"):
            plan.append((wl, "short", None))
            plan.append((wl, "ragged", None))
        for wl, variant, n_wl in plan:
            progress("sub-result %s%s" % (wl, "_" + variant if variant else ""))
            if variant in ("short", "ragged"):
                subs.append(run_lines(wl, variant))
                continue
            r = run(wl, variant, n_wl, with_cpu=(variant is None))
            r.pop("_buf", None)
            dg = r.pop("_digest", None)
            if variant is not None and wl == a.workload:
                same = dg is not None and dg == main_res.get("_digest")
                r["parity_vs_main_run"] = ("bit-exact: sum, index-weighted sum and accept count of all end states equal the main run's" if same else "MISMATCH")
                if not same:
                    r["value"] = None
            r["workload"] = wl + ("_" + variant if variant else "")
            if r["config"].get("table_layout") in ("comb256", "lds", "lds2", "comb"):
                lds_chain_ceiling(r)        # the lookup layouts' own denominator beside the HBM one
            subs.append(r)
        progress("sub-result c3_eager40")
        try:
            subs.append(run_eager40())
            lds_chain_ceiling(subs[-1]) if subs[-1]["config"].get("table_layout") in ("comb256", "lds", "lds2", "comb") else None
        except Exception as e:  # noqa: BLE001
            subs.append({"workload": "c3_eager40", "value": None, "error": repr(e)[:300]})
        progress("sub-result multi_dfa")
        try:
            main_res["multi_dfa"] = run_multi_dfa()
        except Exception as e:  # noqa: BLE001
            main_res["multi_dfa"] = {"error": repr(e)[:300]}
        progress("sub-result multi_dfa_bulk")
        try:
            main_res["multi_dfa_bulk"] = run_multi_dfa_bulk()
        except Exception as e:  # noqa: BLE001
            main_res["multi_dfa_bulk"] = {"error": repr(e)[:300]}
        # leave the main workload's inputs in the buffer for the stream probe below
    if rank != 0:
        if dist_on:
            dist.barrier()   # leave together with rank 0, which still probes the stream rate and prints
            dist.destroy_process_group()
        return

    buf, bm = main_res.pop("_buf")
    main_res.pop("_digest", None)
    stream_gbps = None
    try:  # what a read-only kernel sustains over the same resident bytes (three access patterns, the fastest)
        stream_gbps = hip.stream_read_probe_gbps(buf.data_ptr(), buf.numel(), bm.data_ptr(), 3, stream)
    except Exception:
        pass
    res = {
        "metric": "input GB/s matched (whole node)", "value": main_res["value"], "unit": "GB/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": main_res["ms_per_step"],
        "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": dict(main_res["config"], requested_inputs_per_gpu=requested if a.scaling == "weak" else requested // world),
        "roofline": dict(main_res["roofline"],
                         measured_read_stream_GBps=None if stream_gbps is None else round(stream_gbps, 1),
                         frac_of_measured_stream=None if not stream_gbps else round(main_res["roofline"]["achieved"] / stream_gbps, 4)),
    }
    for k in ("cpu_baseline", "parity_vs_cpu_sample", "parity_sample", "full_parity", "multi_gpu", "multi_dfa", "multi_dfa_bulk"):
        if k in main_res:
            res[k] = main_res[k]
    # N = 1 on a box that shows several GPUs: the C multi-device front on all of them, in a subprocess of its own
    # (its failure must not cost the line above); FSM_BENCH_NODE_FRONT=1 forces it on a one-GPU box
    if not dist_on and a.subs == "auto" and a.n == 0 and (torch.cuda.device_count() > 1 or os.environ.get("FSM_BENCH_NODE_FRONT")):
        import subprocess
        del buf, bm
        buf_all = end_all = None
        torch.cuda.empty_cache()
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--node-front", "--workload", a.workload, "--steps", str(a.steps),
                   "--warmup", str(a.warmup), "--len", str(L)]
            # bounded, but generously: every device first generates its own 100 GB of inputs, and the first RCCL communicator
            # of a process takes tens of seconds to come up
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=600 + 120 * torch.cuda.device_count())
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            res["node_front"] = json.loads(line[-1]) if line else {"error": (out.stderr or out.stdout)[-400:]}
        except Exception as e:  # noqa: BLE001
            res["node_front"] = {"error": repr(e)[:300]}
    if subs:
        res["sub_results"] = subs
        if any(s.get("parity_vs_cpu_sample") == "MISMATCH" for s in subs):
            res["value"] = None
    emit(res)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
