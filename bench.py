#!/usr/bin/env python3
"""bench.py -- the hot path (batched DFA walk) on N GPUs of one node.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c5] [--n INPUTS_PER_GPU]

A "step" is one pass of the walk kernel over one batch of synthetic inputs that
is already resident in HBM (generated on the device, so nothing crosses PCIe),
followed -- for N > 1 -- by the RCCL all-gather of the accept bitmap (asynchronous: it
overlaps the next step's kernel; all K gathers finish inside the timed region).  Inputs are
sharded by contiguous global index range, one shard per rank (weak scaling:
per-GPU work is fixed).  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs):
  c2  configs[1]: PCRE [Ll]ibf+(sm)* DFA (5 states) over 1e8 x 1 KiB random
      inputs per GPU, "Libfsm" planted in every 8th input.        (default)
  c3  configs[2]: 1 024 anchored PCRE unioned into a ~4 096-state DFA, half the
      inputs derived from a pattern (prefix + digits + suffix), half random.
  c5  configs[4]: Aho-Corasick DFA of 1e5 literals (8-16 characters over 64
      symbols, ~1e6 states, right-anchored, end-id = literal), built by the
      library's own fsm_hip_strings_* builder; 1e7 x 1 KiB inputs over the same
      alphabet, every 8th ending with a literal.  The table does not fit LDS: this
      walk is bound by L2 gather requests, not by HBM.
The c2/c3 DFA tables come from tests/golden/{c1,c3}.npz (flattened from the real
reference by tests/golden/make_golden.py); /root/reference is not needed.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
SEED = 0x5EEDF5A1
ALNUM = b"abcdefghijklmnopqrstuvwxyz0123456789"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5"])
    ap.add_argument("--n", "--inputs", dest="n", type=int, default=0, help="inputs per GPU (default 1e8; c5: 1e7)")
    ap.add_argument("--c5-words", type=int, default=100_000, help="c5: number of literals")
    ap.add_argument("--len", type=int, default=1024, help="bytes per input")
    ap.add_argument("--input-mode", type=int, default=-1)
    ap.add_argument("--nb", type=int, default=0)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--mask", type=int, default=-1)
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--layout", type=int, default=0)
    ap.add_argument("--knob", action="append", default=[], help="K=V: raw fsm_hip_dfa_tune knob (see include/fsm_hip_plan.h)")
    ap.add_argument("--no-early-retire", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="inputs for the CPU baseline (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def c3_affixes():
    pats = bytes(np.load(os.path.join(ROOT, "tests", "golden", "c3.npz"))["patterns"]).split(b"\n")
    return [p[1:p.index(b"[")] for p in pats], [b"x", b"yz"]


ALPHA64 = b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_-"
_C5 = {}


def c5_words(nwords):
    """configs[4]: seeded literals of 8-16 characters over a 64-symbol alphabet."""
    if nwords not in _C5:
        rng = np.random.RandomState(SEED & 0x7FFFFFFF)
        alpha = np.frombuffer(ALPHA64, np.uint8)
        _C5[nwords] = [bytes(alpha[rng.randint(0, 64, rng.randint(8, 17))]) for _ in range(nwords)]
    return _C5[nwords]


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
    pf, sf = c3_affixes()
    return hip.gen_affix_inputs_host(n, L, first, SEED, ALNUM, b"0123456789", pf, sf, 2)


def cpu_baseline(hip, workload, flat, L, sample, gpu_end_sample, words=None):
    """Time the reference's CPU path on a bounded sample of the same inputs (rank 0, N=1 only)
    and check the GPU's answers on that sample against it, bit for bit."""
    from oracle import pyoracle
    rows = generate_host(hip, workload, sample, L, 0, words)
    out = {"cores": 1, "unit": "GB/s"}
    gb = rows.size / 1e9
    if workload == "c5":
        # the reference needs ~1 min and 7.5 GB for re_strings on 1e5 literals, 3.7 s per fsm_exec call
        # (fsm_all(isdfa) over 1e6 states, exec.c:106) and 7 min to compile its VM (measured in the build
        # container): the CPU leg of this workload is the oracle's table walker
        o = pyoracle.Oracle(flat)
        want = o.table_walk(rows)
        out.update(kind="port", value=round(gb / o.last_seconds, 5),
                   sample=f"oracle dense-table walker (oracle/dfa_oracle.c), 1 thread, first {sample} inputs x {L} B")
        return out, bool(np.array_equal(gpu_end_sample, want))
    if pyoracle.have_ref():
        # the real reference, rebuilt as a struct fsm from its own regex sources
        if workload == "c2":
            f = pyoracle.RefFsm.re_comp("pcre", b"[Ll]ibf+(sm)*", 0, True, True, endid=0)
            nfe = sample
        else:
            pats = bytes(np.load(os.path.join(ROOT, "tests", "golden", "c3.npz"))["patterns"]).split(b"\n")
            f = pyoracle.RefFsm.union_res("pcre", pats, 0)
            nfe = min(sample, 1500)  # fsm_exec re-runs fsm_all(isdfa) per call: ~9 ms/call on 4k states
        ret, end = f.exec_stride(rows[:nfe])
        t_exec = f.last_seconds
        vm = f.vm_match_stride(rows, 2)
        t_vm = f.last_seconds
        want = pyoracle.Oracle(flat).table_walk(rows)
        assert np.array_equal(end, want[:nfe]), "oracle != reference fsm_exec"
        assert np.array_equal(vm == 1, want != 0xFFFFFFFF), "reference VM != fsm_exec"
        # all host cores: the reference itself is single-threaded, so its fastest matcher (VM v2) is run
        # on one thread per core over slices of the sample, repeated to ~1-2 s of wall time
        ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        quota = None
        try:  # a container may be CPU-throttled far below the visible core count
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = None if q == "max" else round(int(q) / int(per), 2)
        except Exception:
            pass
        if quota:
            ncores = max(1, min(ncores, int(quota + 0.5)))          # threads = cores the cgroup really grants
        f.match_threads(rows, ncores, 1, 2)                         # untimed pass (burst credit, page faults)
        probe, _ = f.match_threads(rows, ncores, 2, 2)              # sizes the timed run (~2 s)
        reps = max(1, min(2000, int(2.0 * probe / gb)))
        allc, acc = f.match_threads(rows, ncores, reps, 2)
        assert acc == int((want != 0xFFFFFFFF).sum()), "threaded VM run disagrees"
        out.update(kind="reference", value=round(nfe * L / 1e9 / t_exec, 5),
                   sample=f"reference fsm_exec (src/libfsm/exec.c) with a (ptr,len) getc, 1 thread, first {nfe} inputs x {L} B of the same generator stream",
                   vm_v2_value=round(gb / t_vm, 5), vm_v2_sample=f"reference fsm_vm_match_buffer v2, 1 thread, {sample} inputs",
                   vm_v2_allcores_value=round(allc, 3), vm_v2_allcores_cores=ncores, vm_v2_allcores_cgroup_cpu_quota=quota,
                   vm_v2_allcores_sample=f"same VM shared read-only by {ncores} threads, each walking its slice of the {sample}-input sample {reps}x")
    else:
        o = pyoracle.Oracle(flat)
        want = o.table_walk(rows)
        out.update(kind="port", value=round(gb / o.last_seconds, 5),
                   sample=f"oracle dense-table walker (oracle/dfa_oracle.c), 1 thread, first {sample} inputs x {L} B")
    parity = bool(np.array_equal(gpu_end_sample, want))
    return out, parity


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU: there is no CPU fallback"
    # FSM_BENCH_BACKEND=gloo lets the N > 1 plumbing be exercised on a box with fewer GPUs than ranks
    # (tests/test_gpu_parity.py::test_bench_two_ranks_share_one_gpu); the real runs use RCCL.
    backend = os.environ.get("FSM_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    import libfsm_amd as hip
    hip.load_library()

    words = None
    if a.workload == "c5":
        words = c5_words(a.c5_words)
        # right-anchored, end-id = literal index: no absorbing accept state, every byte is walked
        flat = hip.FlatDfa.from_strings(words, 2, list(range(len(words))))
    else:
        flat = hip.FlatDfa.load(os.path.join(ROOT, "tests", "golden", "c1.npz" if a.workload == "c2" else "c3.npz"))
    flags = a.layout | (hip.NO_EARLY_RETIRE if a.no_early_retire else 0)
    dfa = hip.HipDfa(flat, flags)
    for knob, v in ((hip.KNOB_INPUT_MODE, a.input_mode), (hip.KNOB_NB, a.nb), (hip.KNOB_ROWS, a.rows),
                    (hip.KNOB_WAVES, a.waves), (hip.KNOB_BLOCKS_PER_CU, a.blocks_per_cu), (hip.KNOB_MASK, a.mask)):
        if v > 0 or (knob in (hip.KNOB_INPUT_MODE, hip.KNOB_MASK) and v >= 0):
            dfa.tune(knob, v)
    for kv in a.knob:
        k, v = kv.split("=")
        dfa.tune(int(k), int(v))
    info = dfa.info()

    L = a.len
    n = a.n if a.n > 0 else (10_000_000 if a.workload == "c5" else 100_000_000)
    requested = n
    free, total = torch.cuda.mem_get_info()
    need = n * (L + 4) + n // 8 + (1 << 30)
    if need > free * 0.92:  # shrink rather than risk an OOM strike; reported in config
        n = int((free * 0.92 - (1 << 30)) // (L + 5)) // 64 * 64
    from libfsm_amd.shard import shard_range
    first, cnt = shard_range(n * world, rank, world)  # weak scaling: the global batch is world x n inputs
    assert cnt == n or n % 64 != 0
    n = cnt
    buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
    end = torch.empty(n, dtype=torch.int32, device="cuda")
    nwords = (n + 63) // 64
    bm = torch.zeros(nwords, dtype=torch.int64, device="cuda")
    gathered = torch.empty(nwords * world, dtype=torch.int64, device="cuda") if world > 1 else None
    generate(hip, a.workload, buf.data_ptr(), n, L, first, words, buf)
    torch.cuda.synchronize()

    stream = torch.cuda.current_stream().cuda_stream
    kernel_ms = []

    # N > 1: the all-gather of step k's bitmap runs on RCCL's stream while step k+1's walk kernel runs
    # (two bitmap / gather buffers); every gather is waited for before the timed region ends.
    bms = [bm, torch.zeros_like(bm)] if world > 1 else [bm]
    gats = [gathered, torch.empty_like(gathered)] if world > 1 else [None]
    pending = [None, None]
    tick = [0]

    def step(record):
        k = tick[0] % len(bms)
        tick[0] += 1
        if world > 1 and pending[k] is not None:
            pending[k].wait()      # stream-level: the gather that last read this bitmap buffer is done
            pending[k] = None
        dfa.exec_batch_device(buf.data_ptr(), L, n, end.data_ptr(), bms[k].data_ptr(), stream=stream)
        if record:
            kernel_ms.append(dfa.last_kernel_ms())  # HIP events on the launch stream, around the walk kernel only
        if world > 1:
            # the match bitmap over RCCL/xGMI (libfsm_amd/shard.py)
            pending[k] = dist.all_gather_into_tensor(gats[k], bms[k], async_op=True)

    def drain():
        for k in range(len(pending)):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    for _ in range(4):  # setup, untimed: the first launches after a long generator kernel run at ramping clocks
        dfa.exec_batch_device(buf.data_ptr(), L, n, end.data_ptr(), bm.data_ptr(), stream=stream)
    for _ in range(a.warmup):
        step(False)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    accepts = int((end != -1).sum().item())
    acc_t = torch.tensor([accepts], dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(acc_t)
    if rank != 0:
        if world > 1:
            dist.barrier()   # leave together with rank 0, which still probes the stream rate and prints
            dist.destroy_process_group()
        return

    stream_gbps = None
    try:  # what a trivially coalesced read-only kernel sustains over the same resident buffer
        stream_gbps = hip.stream_read_probe_gbps(buf.data_ptr(), n * L, bm.data_ptr(), 3, stream)
    except Exception:
        pass
    ms_step = elapsed / a.steps * 1e3
    total_bytes = float(n) * L * world
    value = total_bytes / (elapsed / a.steps) / 1e9
    k_ms = float(np.mean(kernel_ms))
    alg_bytes = float(n) * (L + 4)  # L bytes read + 4 B end state written per input
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    traffic = None
    pj = os.path.join(ROOT, "profiles", f"pmc_{a.workload}.json")
    if os.path.exists(pj):
        try:
            t = json.load(open(pj))
            if int(t.get("n", 0)) == n and int(t.get("len", 0)) == L:
                traffic = t.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    res = {
        "metric": "input GB/s matched (whole node)", "value": round(value, 2), "unit": "GB/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {
            "workload": ("c2: BASELINE configs[1] -- PCRE [Ll]ibf+(sm)* DFA (5 states, absorbing accept), "
                         if a.workload == "c2" else
                         "c3: BASELINE configs[2] -- 1024 anchored PCRE unioned into one %d-state DFA, " % flat.nstates
                         if a.workload == "c3" else
                         "c5: BASELINE configs[4] -- Aho-Corasick DFA of %d literals (%d states, table > LDS), " % (len(words), flat.nstates))
                        + f"{n} x {L} B synthetic inputs per GPU resident in HBM, 64 inputs/wavefront",
            "inputs_per_gpu": n, "input_len": L, "dfa_states": flat.nstates, "byte_classes": info["nclasses"],
            "table_layout": info["layout_name"], "table_bytes": info["table_bytes"], "lds_bytes_per_block": info["lds_bytes"],
            "waves_per_block": info["waves_per_block"], "sharding": f"{world} contiguous index ranges; RCCL all-gather of each step's accept bitmap, overlapped with the next step's walk" if world > 1 else "single GPU",
            "accepted_inputs": int(acc_t.item()), "requested_inputs_per_gpu": requested,
        },
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "measured_read_stream_GBps": None if stream_gbps is None else round(stream_gbps, 1),
                     "frac_of_measured_stream": None if not stream_gbps else round(achieved / stream_gbps, 4),
                     "kernel": "walk (fsmhip::walk_*)", "kernel_ms_avg": round(k_ms, 4),
                     "algorithmic_bytes_per_launch": alg_bytes},
    }
    if a.workload == "c5":
        res["roofline"]["note"] = ("this walk is bound by L2 gather requests, not by HBM: profiles/r01g_c5_rocprof_summary.json "
                                   "(TCC_REQ 6.96e9 per 1e7 x 1 KiB launch = 205 G requests/s, 96.5 % hits; a 4-byte gather test peaks "
                                   "at ~265 G/s, profiles/r01_c5_global_hot.txt); HBM traffic is 3.0x the input bytes")
    if world == 1 and not a.no_cpu_baseline and a.cpu_sample != 0:
        sample = a.cpu_sample if a.cpu_sample > 0 else (400_000 if a.workload == "c2" else 100_000)
        sample = min(sample, n)
        cb, parity = cpu_baseline(hip, a.workload, flat, L, sample, end[:sample].cpu().numpy().view(np.uint32), words)
        res["cpu_baseline"] = cb
        res["parity_vs_cpu_sample"] = "bit-exact" if parity else "MISMATCH"
        if not parity:
            res["value"] = None  # a fast wrong answer is not a result
    print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
